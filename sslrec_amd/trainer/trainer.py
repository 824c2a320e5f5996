"""Training loop of the general-CF scenario; same public surface as the reference's `Trainer`
(trainer/trainer.py:39-196): `create_optimizer`, `train_epoch`, `train` (optional early stop on
the first configured metric), `evaluate`, `test`, `save_model`, `load_model`; per step
zero_grad -> cal_loss -> backward -> step.  The eleven model-specific trainers upstream belong to
models outside this path and are not provided."""
import os
import time
from copy import deepcopy

import numpy as np
import torch
import torch.optim as optim

from ..config.configurator import configs
from ..models.bulid_model import build_model
from .metrics import Metric
from .utils import DisabledSummaryWriter, log_exceptions

writer = DisabledSummaryWriter()


def init_seed():
    """Seed numpy (negative sampling), torch CPU (init, shuffling, augmentation draws) and the GPU
    generators when train.reproducible is set (reference :26-36)."""
    train_cfg = configs['train']
    if train_cfg.get('reproducible'):
        seed = train_cfg['seed']
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)


class Trainer(object):
    def __init__(self, data_handler, logger):
        self.data_handler = data_handler
        self.logger = logger
        self.metric = Metric()

    def create_optimizer(self, model):
        cfg = configs['optimizer']
        if cfg['name'] == 'adam':
            self.optimizer = optim.Adam(model.parameters(), lr=cfg['lr'], weight_decay=cfg['weight_decay'])
        else:
            raise NotImplementedError("optimizer '%s'" % cfg['name'])

    def train_epoch(self, model, epoch_idx):
        loader = self.data_handler.train_dataloader
        loader.dataset.sample_negs()
        loss_log = {}
        ep_loss = 0.0
        n_batches = len(loader)
        model.train()
        for tem in loader:
            self.optimizer.zero_grad()
            batch_data = [x.long().to(configs['device']) for x in tem]
            loss, loss_dict = model.cal_loss(batch_data)
            ep_loss += loss.item()
            loss.backward()
            self.optimizer.step()
            for name, value in loss_dict.items():
                value = value.item() if torch.is_tensor(value) else float(value)
                loss_log[name] = loss_log.get(name, 0.0) + value / n_batches
        steps = max(1, len(loader.dataset) // configs['train']['batch_size'])
        writer.add_scalar('Loss/train', ep_loss / steps, epoch_idx)
        self.logger.log_loss(epoch_idx, loss_log, save_to_log=bool(configs['train']['log_loss']))

    @log_exceptions
    def train(self, model):
        self.create_optimizer(model)
        cfg = configs['train']
        if not cfg['early_stop']:
            for epoch_idx in range(cfg['epoch']):
                self.train_epoch(model, epoch_idx)
                if epoch_idx % cfg['test_step'] == 0:
                    self.evaluate(model, epoch_idx)
            self.test(model)
            self.save_model(model)
            return model
        patience, best_epoch, best_metric, best_state = 0, 0, -1e9, None
        first_metric = configs['test']['metrics'][0]
        for epoch_idx in range(cfg['epoch']):
            self.train_epoch(model, epoch_idx)
            if epoch_idx % cfg['test_step'] == 0:
                score = self.evaluate(model, epoch_idx)[first_metric][0]
                if score > best_metric:
                    patience, best_epoch, best_metric = 0, epoch_idx, score
                    best_state = deepcopy(model.state_dict())
                    self.logger.log('Validation score increased.  Copying the best model ...')
                else:
                    patience += 1
                    self.logger.log('Early stop counter: {} out of {}'.format(patience, cfg['patience']))
                if patience == cfg['patience']:
                    break
        self.logger.log('Best Epoch {}'.format(best_epoch))
        model = build_model(self.data_handler).to(configs['device'])
        if best_state is not None:
            model.load_state_dict(best_state)
        self.evaluate(model)
        self.test(model)
        self.save_model(model)
        return model

    @log_exceptions
    def evaluate(self, model, epoch_idx=None):
        model.eval()
        if hasattr(self.data_handler, 'valid_dataloader'):
            loader, name = self.data_handler.valid_dataloader, 'Validation set'
        elif hasattr(self.data_handler, 'test_dataloader'):
            loader, name = self.data_handler.test_dataloader, 'Test set'
        else:
            raise NotImplementedError
        result = self.metric.eval(model, loader)
        self.logger.log_eval(result, configs['test']['k'], data_type=name, epoch_idx=epoch_idx)
        return result

    @log_exceptions
    def test(self, model):
        model.eval()
        if not hasattr(self.data_handler, 'test_dataloader'):
            raise NotImplementedError
        result = self.metric.eval(model, self.data_handler.test_dataloader)
        self.logger.log_eval(result, configs['test']['k'], data_type='Test set')
        return result

    def save_model(self, model):
        if not configs['train']['save_model']:
            return
        name, data = configs['model']['name'], configs['data']['name']
        if configs['tune']['enable']:
            directory, tag = './checkpoint/{}/tune'.format(name), configs['tune']['now_para_str']
        else:
            directory, tag = './checkpoint/{}'.format(name), int(time.time())
        os.makedirs(directory, exist_ok=True)
        path = '{}/{}-{}-{}.pth'.format(directory, name, data, tag)
        torch.save(model.state_dict(), path)
        self.logger.log('Save model parameters to {}'.format(path))

    def load_model(self, model):
        if 'pretrain_path' not in configs['train']:
            raise KeyError("No pretrain_path in configs['train']")
        path = configs['train']['pretrain_path']
        model.load_state_dict(torch.load(path))
        self.logger.log('Load model parameters from {}'.format(path))
        return model
