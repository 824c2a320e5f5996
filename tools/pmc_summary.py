#!/usr/bin/env python
"""Turns the rocprofv3 --pmc passes of tools/gpu_profile.sh into profiles/spmm_traffic.json (the `traffic`
field of bench.py's roofline) and a per-kernel counter summary.
usage: python tools/pmc_summary.py gpurun_out/<tag> [profiles/<round>/spmm_pmc_summary.json] [round] [commit]"""
import collections, csv, glob, json, os, subprocess, sys

src = sys.argv[1]
out_round = sys.argv[2] if len(sys.argv) > 2 else None
round_tag = sys.argv[3] if len(sys.argv) > 3 else None
commit = sys.argv[4] if len(sys.argv) > 4 else subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, 'pmc_*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'spmm_' not in name:
            continue
        key = name.split('(')[0].replace('void ', '').strip()
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {}
for k, cs in acc.items():
    summary[k] = {c: {'launches': len(v), 'mean': sum(v) / len(v)} for c, v in cs.items()}
print(json.dumps(summary, indent=1))
main = max((k for k in summary if 'reduce' not in k), key=lambda k: summary[k].get('FETCH_SIZE', {}).get('mean', 0), default=None)
# round 5: a layer chain is one valued launch + pattern launches -- two instantiations of the same kernel template (last template argument):
# the per-launch figure of the step is their launch-weighted mean
family = [k for k in summary if main and k.rsplit(',', 1)[0] == main.rsplit(',', 1)[0] and 'FETCH_SIZE' in summary[k]]
if main and len(family) > 1:
    merged = {}
    for c in set().union(*(summary[k].keys() for k in family)):
        ks = [k for k in family if c in summary[k]]
        n = sum(summary[k][c]['launches'] for k in ks)
        merged[c] = {'launches': n, 'mean': sum(summary[k][c]['mean'] * summary[k][c]['launches'] for k in ks) / n}
    summary[' + '.join(sorted(family))] = merged
    main = ' + '.join(sorted(family))
if main and 'FETCH_SIZE' in summary[main]:
    s = summary[main]
    fetch = s['FETCH_SIZE']['mean'] * 1024 * 2            # KB -> B, doubled: gfx950 tallies 128-B requests at 64 B
    write = s.get('WRITE_SIZE', {}).get('mean', 0.0) * 1024
    hit, miss = s.get('TCC_HIT_sum', {}).get('mean'), s.get('TCC_MISS_sum', {}).get('mean')
    traffic = {
        'kernel': main,
        'measured_in_round': round_tag, 'measured_at_commit': commit,
        'command': 'rocprofv3 --pmc <C> --kernel-trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras '
                   '(separate passes for FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum+TCC_MISS_sum; tools/gpu_profile.sh)',
        'FETCH_SIZE_KB_raw': s['FETCH_SIZE']['mean'], 'WRITE_SIZE_KB_raw': s.get('WRITE_SIZE', {}).get('mean'),
        'correction': 'FETCH_SIZE doubled (gfx950 counts 128-B fabric requests at 64 B); WRITE_SIZE used as reported (uncalibrated)',
        'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write,
        'TCC_HIT_sum': hit, 'TCC_MISS_sum': miss, 'l2_hit_rate': (hit / (hit + miss)) if hit is not None and miss else None,
        'note': 'per launch of the dominant SpMM kernel inside the bench step (fused-accumulator launches); the 37 MB operand fits the '
                '256 MiB Infinity Cache, so most of these fabric reads are served on-die, not by HBM; the counter sits on the L2 memory side',
    }
    json.dump(traffic, open(os.path.join(ROOT, 'profiles', 'spmm_traffic.json'), 'w'), indent=1)
    print('wrote profiles/spmm_traffic.json:', main, '%.0f MB per launch' % ((fetch + write) / 1e6))
if out_round:
    json.dump(summary, open(out_round, 'w'), indent=1)
