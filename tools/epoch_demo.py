#!/usr/bin/env python
"""End-to-end demo at BASELINE cfg-2 scale: data handler -> model -> Trainer for a few epochs on the
amazon-book-shaped synthetic graph, perf-mode switches on (device RNG, vectorized negative sampling,
device-side evaluation mask).  usage: python tools/epoch_demo.py [model] [epochs] [graph] [fused] [x36]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslrec_amd.config.configurator import load_config
model_name = sys.argv[1] if len(sys.argv) > 1 else 'lightgcn'
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hip_graph = 'graph' in sys.argv[3:]
os.makedirs('/tmp/sslrec_demo', exist_ok=True); os.chdir('/tmp/sslrec_demo')
load_config(model_name, device='cuda', overrides={
    'data': {'synthetic': 'amazon-book'},
    'train': {'epoch': epochs, 'test_step': 1, 'fast_neg_sampling': True, 'fast_loader': True, 'device_sampler': True, 'patience': 10, 'hip_graph': hip_graph},
    'model': dict({'embedding_size': 64, 'layer_num': 3, 'device_rng': True},
                  **({'infonce_precision': 'x36'} if 'x36' in sys.argv[3:] else {})),
    'optimizer': {'fused': 'fused' in sys.argv[3:]}})
from sslrec_amd.data_utils.build_data_handler import build_data_handler
from sslrec_amd.models.bulid_model import build_model
from sslrec_amd.trainer.build_trainer import build_trainer
from sslrec_amd.trainer.logger import Logger
from sslrec_amd.trainer.trainer import init_seed
t0 = time.time(); init_seed(); dh = build_data_handler(); dh.load_data(); print('load_data %.1f s' % (time.time() - t0))
model = build_model(dh).to('cuda'); trainer = build_trainer(dh, Logger(log_configs=False)); trainer.create_optimizer(model)
for ep in range(epochs):
    torch.cuda.synchronize(); t1 = time.time()
    trainer.train_epoch(model, ep)
    torch.cuda.synchronize(); t2 = time.time()
    trainer.evaluate(model, ep)
    torch.cuda.synchronize(); t3 = time.time()
    print('epoch %d: train %.2f s (%d steps, %.2f ms/step incl. data loader), eval %.2f s' % (
        ep, t2 - t1, len(dh.train_dataloader), (t2 - t1) / len(dh.train_dataloader) * 1e3, t3 - t2))
