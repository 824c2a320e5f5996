"""File + console logger with the reference's message formats (trainer/logger.py:10-59):
`./log/{model}/{dataset}_{time}.log`, one line per epoch of losses, one per evaluation."""
import datetime
import logging
import os

from ..config.configurator import configs


def get_local_time():
    return datetime.datetime.now().strftime('%b-%d-%Y_%H-%M-%S')


class Logger(object):
    def __init__(self, log_configs=True):
        model_name = configs['model']['name']
        log_dir = './log/{}'.format(model_name)
        os.makedirs(log_dir, exist_ok=True)
        self.logger = logging.getLogger('train_logger')
        self.logger.setLevel(logging.INFO)
        suffix = '-tune' if configs['tune']['enable'] else ''
        handler = logging.FileHandler('{}/{}{}_{}.log'.format(log_dir, configs['data']['name'], suffix, get_local_time()),
                                      'a', encoding='utf-8')
        handler.setFormatter(logging.Formatter('%(asctime)s - %(message)s'))
        self.logger.addHandler(handler)
        if log_configs:
            self.log(configs)

    def log(self, message, save_to_log=True, print_to_console=True):
        if save_to_log:
            self.logger.info(message)
        if print_to_console:
            print(message)

    def log_loss(self, epoch_idx, loss_log_dict, save_to_log=True, print_to_console=True):
        message = '[Epoch {:3d} / {:3d}] '.format(epoch_idx, configs['train']['epoch'])
        for name, value in loss_log_dict.items():
            message += '{}: {:.4f} '.format(name, value)
        self.log(message, save_to_log, print_to_console)

    def log_eval(self, eval_result, k, data_type, save_to_log=True, print_to_console=True, epoch_idx=None):
        message = '' if epoch_idx is None else 'Epoch {:3d} '.format(epoch_idx)
        for metric, values in eval_result.items():
            message += '{} ['.format(data_type)
            for i, kk in enumerate(k):
                message += '{}@{}: {:.4f} '.format(metric, kk, values[i])
            message += '] '
        self.log(message, save_to_log, print_to_console)
