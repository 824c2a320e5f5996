"""Model factory: class looked up by reflection in `models/{data.type}/{model.name}.py`, matched
case-insensitively on the class name (behaviour of the reference's models/bulid_model.py:4-15;
the module keeps the upstream file-name spelling so imports port unchanged)."""
import importlib
import inspect

from ..config.configurator import configs


def _find_class(module, wanted):
    for name, obj in inspect.getmembers(module, inspect.isclass):
        if name.lower() == wanted and obj.__module__ == module.__name__:
            return obj
    return None


def build_model(data_handler):
    scenario, wanted = configs['data']['type'], configs['model']['name'].lower()
    module_path = '{}.{}.{}'.format(__package__, scenario, wanted)
    try:
        module = importlib.import_module(module_path)
    except ModuleNotFoundError as exc:
        if exc.name and not module_path.startswith(exc.name) and exc.name != module_path:
            raise                                  # a genuine missing dependency inside the model file
        raise NotImplementedError('Model {} is not implemented'.format(wanted)) from exc
    cls = _find_class(module, wanted)
    if cls is None:
        raise NotImplementedError('Model Class {} is not defined in {}'.format(wanted, module_path))
    return cls(data_handler)
