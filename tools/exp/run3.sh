for rep in 1 2; do for v in Y1 Z0; do cp ab/libvali_hip_$v.so vali_amd/libvali_hip.so; echo "== $v"; python tools/bench_configs.py interp 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        for r in d.get('results',[]):
            if 'ws' in r['kernel']: print(r['format'], r['geometry'], r['kernel'][:28], r['us_per_frame'], r['roofline']['frac'])
"; done; done; cp ab/libvali_hip_Y1.so vali_amd/libvali_hip.so
