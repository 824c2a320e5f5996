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
            # train.hip_graph replays the step from a captured hipGraph: the step counter must live on the device
            graphed = bool(configs['train'].get('hip_graph'))
            if cfg.get('fused'):        # one HIP pass per table, step counter on the device (sslrec_amd/optim.py)
                from ..optim import FusedAdam
                self.optimizer = FusedAdam(model.parameters(), lr=cfg['lr'], weight_decay=cfg['weight_decay'])
            else:
                self.optimizer = optim.Adam(model.parameters(), lr=cfg['lr'], weight_decay=cfg['weight_decay'],
                                            **({'capturable': True} if graphed else {}))
        else:
            raise NotImplementedError("optimizer '%s'" % cfg['name'])
        self._graph = None

    # ---- opt-in: one training step captured as a hipGraph (train.hip_graph: true) ------------------------
    # A LightGCN step is ~40 kernel launches of 5-100 us; issued from Python they take ~1.4 ms of wall time for
    # ~0.9 ms of GPU work.  The step (cal_loss, backward, Adam) is captured ONCE after a few eager warm-up steps
    # and replayed with the batch copied into static index tensors; losses are accumulated on the device and read
    # once per epoch.  Needs device-side augmentation RNG (model.device_rng) -- a host draw cannot be captured --
    # and full batches (the last, shorter batch of an epoch runs eagerly).
    def _backward(self, loss):
        """loss.backward() with d loss / d loss = 1 made once per device instead of filled by autograd on every step"""
        key = (loss.device, loss.dtype)
        one = self._ones.get(key) if hasattr(self, '_ones') else None
        if one is None:
            if not hasattr(self, '_ones'):
                self._ones = {}
            one = self._ones[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(one if loss.dim() == 0 else None)

    def _eager_step(self, model, batch_data):
        self.optimizer.zero_grad(set_to_none=True)
        loss, loss_dict = model.cal_loss(batch_data)
        self._backward(loss)
        self.optimizer.step()
        return loss, loss_dict

    @staticmethod
    def _device_of(model):
        """device of the model's parameters; a plugin that is not an nn.Module, or has no parameters, runs on the configured device"""
        try:
            return next(model.parameters()).device
        except (AttributeError, StopIteration, TypeError):
            return torch.device(configs['device'])

    @staticmethod
    def _host_rng_in_step(model):
        """True when the model's training forward draws from the CPU generator (the reference's EdgeDrop / EmbedPerturb
        behaviour, kept for bit parity unless model.device_rng is set): such a draw + H2D copy cannot be captured"""
        mcfg = configs['model']
        if mcfg.get('device_rng'):
            return False
        from ..rng import active_host_replay
        dev = Trainer._device_of(model)
        if active_host_replay(dev) is not None:
            return False         # the CPU generator's stream is produced by a kernel (sslrec_amd/csrc/mt19937.hip): capturable
        name = type(model).__name__.lower()
        return name in ('sgl', 'simgcl') or (name == 'lightgcn' and float(mcfg.get('keep_rate', 1.0)) != 1.0)

    def _capture_step(self, model, batch_data):
        if self._host_rng_in_step(model):
            raise RuntimeError('train.hip_graph needs device-side augmentation RNG: set model.device_rng (the CPU draws of '
                               'EdgeDrop / EmbedPerturb and their host-to-device copies cannot be captured in a hipGraph)')
        dev = batch_data[0].device
        static_batch = [b.clone() for b in batch_data]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # warm-up (layouts, workspaces, optimizer state, LDS attributes) that must NOT advance the training:
            # parameters, moments and step counters are put back IN PLACE afterwards -- the captured graph holds
            # their addresses.  Optimizer state that did not exist before the warm-up goes back to zero.
            def opt_tensors():
                from ..rng import active_host_replay
                replay = active_host_replay(dev)
                if replay is not None:
                    replay.attach()                                  # its state is on the device before the snapshot is taken
                out = [replay.mt] if replay is not None else []      # the warm-up's draws must not advance the replayed generator
                out += list(getattr(self.optimizer, '_ticks', {}).values())
                for st in self.optimizer.state.values():
                    out += [v for v in st.values() if torch.is_tensor(v)]
                return out
            params = [p.data for p in model.parameters()]
            before = {id(t): t.clone() for t in params + opt_tensors()}
            for _ in range(3):
                self._eager_step(model, static_batch)
            for t in params + opt_tensors():
                if id(t) in before:
                    t.copy_(before[id(t)])
                else:
                    t.zero_()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss, loss_dict = model.cal_loss(static_batch)
            self._backward(loss)
            self.optimizer.step()
            outs = {'loss': loss.detach().float().reshape(())}
            for name, value in loss_dict.items():
                outs['part_' + name] = (value.detach() if torch.is_tensor(value) else torch.tensor(float(value), device=dev)).float().reshape(())
        return {'graph': graph, 'batch': static_batch, 'outs': outs, 'B': int(batch_data[0].shape[0])}

    def _train_epoch_graphed(self, model, epoch_idx):
        loader = self.data_handler.train_dataloader
        loader.dataset.sample_negs()
        dev = configs['device']
        model.train()
        sums, n_batches = {}, len(loader)

        def add(outs):
            for k, v in outs.items():
                sums[k] = sums[k] + v if k in sums else v.clone()

        from ..rng import active_host_replay
        replay = None if configs['model'].get('device_rng') else active_host_replay(self._device_of(model))
        attached = False
        for tem in loader:
            batch_data = [x.long().to(dev) for x in tem]
            st = self._graph
            if st is None and batch_data[0].shape[0] == configs['train']['batch_size']:
                st = self._graph = self._capture_step(model, batch_data)
                # the warm-up steps' effect on parameters and optimizer state was undone in place and capture executes
                # nothing: the replay below is this batch's one real step
            if st is not None and batch_data[0].shape[0] == st['B']:
                if replay is not None and not attached:
                    # a replay bypasses HostGeneratorReplay.rand()/keep_mask(): follow the host generator here if it moved
                    # since the last flush (the DataLoader's shuffle seed is drawn from it when the iterator is created),
                    # or raise if it moved while the device was ahead
                    replay.attach()
                    attached = True
                for dst, src in zip(st['batch'], batch_data):
                    dst.copy_(src)
                st['graph'].replay()
                if replay is not None:
                    replay.ahead = True      # the captured generator kernels advanced the device state: flush() must write it back
                add(st['outs'])
            else:
                loss, loss_dict = self._eager_step(model, batch_data)
                outs = {'loss': loss.detach().float().reshape(())}
                for name, value in loss_dict.items():
                    outs['part_' + name] = (value.detach() if torch.is_tensor(value) else torch.tensor(float(value), device=dev)).float().reshape(())
                add(outs)
        host = {k: float(v.item()) for k, v in sums.items()}              # one synchronisation per epoch
        steps = max(1, len(loader.dataset) // configs['train']['batch_size'])
        writer.add_scalar('Loss/train', host.get('loss', 0.0) / steps, epoch_idx)
        loss_log = {k[5:]: v / n_batches for k, v in host.items() if k.startswith('part_')}
        self.logger.log_loss(epoch_idx, loss_log, save_to_log=bool(configs['train']['log_loss']))

    def train_epoch(self, model, epoch_idx):
        try:
            if configs['train'].get('hip_graph'):
                return self._train_epoch_graphed(model, epoch_idx)
            return self._train_epoch_eager(model, epoch_idx)
        finally:
            # parity-mode draws are produced on the device (sslrec_amd.rng.HostGeneratorReplay): hand the generator back to
            # the host before anything there draws again (next epoch's shuffling, evaluation, model construction)
            from ..rng import flush_host_replay
            flush_host_replay()

    def _train_epoch_eager(self, model, epoch_idx):
        """reference trainer/trainer.py:51-84.  The reference reads `loss.item()` (and every logged part) in every step: three device
        synchronisations per step, during which the host cannot queue the next step's launches.  Here the step's 0-d tensors are kept and
        read ONCE at the end of the epoch; the sums are then formed on the host in the reference's order and arithmetic (Python floats,
        `value / n_batches` per step), so the logged numbers are the same -- only their moment of transfer moved."""
        loader = self.data_handler.train_dataloader
        loader.dataset.sample_negs()
        n_batches = len(loader)
        model.train()
        kept, names = [], None
        for tem in loader:
            self.optimizer.zero_grad()
            batch_data = [x.long().to(configs['device']) for x in tem]
            loss, loss_dict = model.cal_loss(batch_data)
            self._backward(loss)
            self.optimizer.step()
            if names is None:
                names = list(loss_dict)
            dev = loss.device
            kept.append([loss.detach().reshape(())] + [(loss_dict[k].detach().reshape(()).to(loss.dtype) if torch.is_tensor(loss_dict[k])
                                                        else torch.tensor(float(loss_dict[k]), dtype=loss.dtype, device=dev)) for k in names])
        loss_log, ep_loss = {}, 0.0
        if kept:
            host = torch.stack([torch.stack(row) for row in kept]).double().cpu().tolist()      # one synchronisation per epoch
            for row in host:
                ep_loss += row[0]
                for name, value in zip(names, row[1:]):
                    loss_log[name] = loss_log.get(name, 0.0) + value / n_batches
        steps = max(1, len(loader.dataset) // configs['train']['batch_size'])
        writer.add_scalar('Loss/train', ep_loss / steps, epoch_idx)
        self.logger.log_loss(epoch_idx, loss_log, save_to_log=bool(configs['train']['log_loss']))

    @log_exceptions
    def train(self, model):
        from .. import rng
        dev = self._device_of(model)
        # train.host_rng_replay (default on): the reference's CPU draws for EdgeDrop / EmbedPerturb come out of the same
        # generator algorithm running on the GPU -- same numbers, no host stall; flushed back at every epoch boundary
        replay = dev.type == 'cuda' and configs['train'].get('host_rng_replay', True) and not configs['model'].get('device_rng')
        mine = replay and rng.active_host_replay(dev) is None      # a replay the caller enabled stays the caller's
        rep, ahead_before = None, None
        if replay:
            rep = rng.enable_host_replay(dev)
            ahead_before = rep.draw_ahead
            if configs['train'].get('hip_graph'):      # a captured step generates its numbers inside the graph: nothing to draw ahead
                rep.draw_ahead = False
        try:
            return self._train(model)
        finally:
            if mine:
                rng.disable_host_replay(dev)
            elif replay:
                rep.draw_ahead = ahead_before          # the caller's replay goes back as it came
                rng.flush_host_replay()

    def _train(self, model):
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
