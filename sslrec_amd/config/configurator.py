"""Global `configs` dictionary, same keys and derivations as the reference's
config/configurator.py:5-57 (YAML `config/modelconf/{model}.yml` + the four CLI flags
--model/--dataset/--device/--cuda; derived keys tune.enable, device, train.log_loss,
train.early_stop).

Difference in mechanics only: the reference parses sys.argv as an import side effect;
here the dict object is created empty at import and filled either by `parse_configure()`
(CLI entry, main.py) or by `load_config(...)` (library / tests / bench).  Every module
keeps a reference to the SAME dict object, so the reference's habit of mutating
`configs` at run time (data handler, tuner) keeps working.
"""
import argparse
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
MODELCONF_DIR = os.path.join(_HERE, 'modelconf')

configs = {}


def _derive(cfg, device, dataset):
    cfg['model']['name'] = cfg['model']['name'].lower()
    if 'tune' not in cfg:
        cfg['tune'] = {'enable': False}
    cfg['device'] = device
    if dataset is not None:
        cfg['data']['name'] = dataset
    if 'log_loss' not in cfg['train']:
        cfg['train']['log_loss'] = True
    if 'patience' in cfg['train']:
        if cfg['train']['patience'] <= 0:
            raise Exception("'patience' should be greater than 0.")
        cfg['train']['early_stop'] = True
    else:
        cfg['train']['early_stop'] = False
    return cfg


def _find_yml(model_name):
    for base in ('./config/modelconf', MODELCONF_DIR):      # cwd-relative first, like the reference
        path = os.path.join(base, '{}.yml'.format(model_name))
        if os.path.exists(path):
            return path
    raise Exception('Please create the yaml file for your model first.')


def load_config(model, dataset=None, device='cuda', overrides=None):
    """Fill the global dict for `model`; `overrides` = {'model': {...}, 'train': {...}, ...}
    merged on top of the YAML (the mechanism trainer/tuner.py:37 uses upstream)."""
    if model is None:
        raise Exception('Please provide the model name through --model.')
    with open(_find_yml(model.lower()), encoding='utf-8') as f:
        cfg = yaml.safe_load(f.read())
    cfg = _derive(cfg, device, dataset)
    for section, kv in (overrides or {}).items():
        cfg.setdefault(section, {}).update(kv)
    configs.clear()
    configs.update(cfg)
    return configs


def parse_configure(argv=None):
    parser = argparse.ArgumentParser(description='SSLRec')
    parser.add_argument('--model', type=str, help='Model name')
    parser.add_argument('--dataset', type=str, default=None, help='Dataset name')
    parser.add_argument('--device', type=str, default='cuda', help='cpu or cuda')
    parser.add_argument('--cuda', type=str, default='0', help='Device number')
    args = parser.parse_args(argv)
    if args.device == 'cuda':
        os.environ['CUDA_VISIBLE_DEVICES'] = args.cuda
    return load_config(args.model, args.dataset, args.device)
