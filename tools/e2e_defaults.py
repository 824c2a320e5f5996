#!/usr/bin/env python
"""End-to-end check of the DEFAULT path (every switch at its default: the reference's batches, draws and arithmetic): data handler ->
model -> Trainer.train (epochs, evaluation every epoch, test) for the four models on a synthetic graph of a BASELINE shape.
usage: python tools/e2e_defaults.py [graph=yelp] [epochs=2] [models=lightgcn,sgl,simgcl,lightgcl]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslrec_amd.config.configurator import load_config
graph = sys.argv[1] if len(sys.argv) > 1 else 'yelp'
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
models = (sys.argv[3] if len(sys.argv) > 3 else 'lightgcn,sgl,simgcl,lightgcl').split(',')
os.makedirs('/tmp/sslrec_e2e', exist_ok=True); os.chdir('/tmp/sslrec_e2e')
out = {}
for name in models:
    load_config(name, device='cuda', overrides={'data': {'synthetic': graph}, 'train': {'epoch': epochs, 'test_step': 1, 'early_stop': False, 'save_model': False, 'log_loss': True},
                                                'model': {'embedding_size': 64}})
    from sslrec_amd.data_utils.build_data_handler import build_data_handler
    from sslrec_amd.models.bulid_model import build_model
    from sslrec_amd.trainer.build_trainer import build_trainer
    from sslrec_amd.trainer.logger import Logger
    from sslrec_amd.trainer.trainer import init_seed
    t0 = time.time(); init_seed(); dh = build_data_handler(); dh.load_data(); t_load = time.time() - t0
    model = build_model(dh).to('cuda')
    trainer = build_trainer(dh, Logger(log_configs=False))
    t1 = time.time(); trainer.train(model); torch.cuda.synchronize(); t_train = time.time() - t1
    res = trainer.test(model)
    out[name] = {'load_s': round(t_load, 2), 'train_s_incl_eval': round(t_train, 2), 'test': {k: [round(float(x), 5) for x in v] for k, v in res.items()},
                 'finite': bool(all(torch.isfinite(p).all() for p in model.parameters()))}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
