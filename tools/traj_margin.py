#!/usr/bin/env python
"""How far the 90-step real-yelp trajectories (tests/test_gpu_parity.py::test_training_trajectory_on_real_yelp_matches_the_reference_run)
end from the reference's final embeddings: max |difference| over the sampled rows, per model -- the margin under the north star's 1e-5.
usage: [SSLREC_INFONCE_BSPLIT=1] python tools/traj_margin.py [sgl lightgcn]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
from sslrec_amd.models.bulid_model import build_model
DEV = 'cuda'
out = {'SSLREC_INFONCE_BSPLIT': os.environ.get('SSLREC_INFONCE_BSPLIT')}
for model_name in (sys.argv[1:] or ['sgl']):
    g, cfg, opt_cfg, meta = H.load_trajectory(model_name, 64, 2, case='yelp')
    dh = H.trajectory_setup(model_name, g, cfg, opt_cfg, meta, DEV)
    model = build_model(dh).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])
    losses = []
    for _ in range(meta['epochs']):
        dh.train_dataloader.dataset.sample_negs()
        for tem in dh.train_dataloader:
            batch = [x.long().to(DEV) for x in tem]
            opt.zero_grad()
            loss, _ = model.cal_loss(batch)
            loss.backward()
            opt.step()
            losses.append(loss.item())
    rec = {'loss_max_rel': float(np.max(np.abs(np.array(losses) - g['losses']) / np.abs(g['losses'])))}
    for name in ('user_embeds', 'item_embeds'):
        got = getattr(model, name).detach().cpu().numpy()[::97]
        d = np.abs(got - g['finalrows_' + name])
        rec[name + '_max_abs'] = float(d.max())
        rec[name + '_over_5e-6'] = int((d > 5e-6).sum())
        rec[name + '_p999'] = float(np.quantile(d, 0.999))
    out[model_name] = rec
print(json.dumps(out), flush=True)
