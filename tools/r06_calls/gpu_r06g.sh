#!/bin/bash
# round 6, call g: h3 on the un-normalized variant (device-chosen plane scales, per-anchor exponent bias): parity against float64 and x6,
# the sharded / LightGCL suites with it as the variant's default, and its time against x6 on config 5's rank-sized term
O=gpurun_out/r06g; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "unnormalized or lightgcl or LightGCL or sharded or two_ranks or infonce" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log | cut -c1-300
timeout 600 python - > $O/v1_times.log 2>&1 <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from sslrec_amd import ops
dev = 'cuda:0'
out = {}
for (M, d, B, temp) in ((91599, 64, 4096, 0.2), (1250000, 128, 4096, 0.5)):
    gen = torch.Generator().manual_seed(1)
    t1 = (torch.randn(M, d, generator=gen) * 0.1).to(dev).requires_grad_(True)
    t2 = (torch.randn(M, d, generator=gen) * 0.1).to(dev).requires_grad_(True)
    idx = torch.randint(0, M, (B,), generator=gen).to(dev)
    rec = {}
    for prec in ('x6', 'h3'):
        def fb():
            t1.grad = t2.grad = None
            ops.infonce_loss_gathered(t1, t2, idx, temp, variant=1, precision=prec).backward()
        def fwd():
            with torch.no_grad():
                return ops.infonce_loss_gathered(t1, t2, idx, temp, variant=1, precision=prec)
        for nm, fn in (('fwd_nograd_ms', fwd), ('fwdbwd_ms', fb)):
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            rec['%s_%s' % (prec, nm)] = e0.elapsed_time(e1) / 5
        rec[prec + '_loss'] = fwd().item()
    out['M=%d d=%d B=%d' % (M, d, B)] = rec
    print(json.dumps({('M=%d d=%d' % (M, d)): rec}), flush=True)
    del t1, t2
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join('gpurun_out/r06g', 'infonce_v1_h3_vs_x6.json'), 'w'), indent=1)
PY
echo "v1 times rc $?"; cat $O/v1_times.log | tail -4 | cut -c1-600
