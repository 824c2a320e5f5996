"""Trainer factory (reference trainer/build_trainer.py:4-13).  The reference picks a
model-specific trainer class when `configs['train']['trainer']` names one; all four models of
this path use the generic `Trainer`."""
from ..config.configurator import configs
from .trainer import Trainer


def build_trainer(data_handler, logger):
    name = configs['train'].get('trainer')
    if name is not None and name.lower() != 'trainer':
        raise NotImplementedError('Trainer {} is not implemented for the general-CF hot path'.format(name))
    return Trainer(data_handler, logger)
