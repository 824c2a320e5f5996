"""Reflection factory: data handler class chosen from configs['data']['type']
(reference data_utils/build_data_handler.py:4-14).  Only general_cf is in scope."""
import importlib
import importlib.util

from ..config.configurator import configs


def build_data_handler():
    name = 'data_handler_' + configs['data']['type']
    module_path = '.'.join([__package__, name])
    if importlib.util.find_spec(module_path) is None:
        raise NotImplementedError('DataHandler {} is not implemented'.format(name))
    module = importlib.import_module(module_path)
    wanted = name.lower().replace('_', '')
    for attr in dir(module):
        if attr.lower() == wanted:
            return getattr(module, attr)()
    raise NotImplementedError('DataHandler Class {} is not defined in {}'.format(name, module_path))
