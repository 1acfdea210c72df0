#!/bin/bash
# dual-launch threshold of the kernels other than k_k3: 1024 (product) vs 3072, same box, alternating; bridge descriptors at both sizes + bench configs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5z; mkdir -p $O
A="$PWD/reference-seal-backend_amd/lib/libhebench_mi355x_backend.so"
B="$PWD/reference-seal-backend_amd/lib/alt_d3072.so"
for rep in 1 2; do for lib in $A $B; do
  HE355_LIB_PATH=$lib python3 tools/bench_bridge.py --sizes both --reps 20 --no-direct --out $O/br_$(basename $lib .so)_$rep.jsonl > /dev/null 2>&1
done; done
python3 - <<'PY'
import json,glob,collections
t=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/r5z/br_*.jsonl')):
    tag=f.split('/')[-1][3:-6]
    for l in open(f):
        r=json.loads(l)
        if r.get('pool','on')!='on': continue
        t[(r['descriptor'],r['sizes'])].setdefault(tag.rsplit('_',1)[0],[]).append(r['operate_ms'])
for k,v in t.items():
    a=v.get('libhebench_mi355x_backend',[]); b=v.get('alt_d3072',[])
    if a and b: print(f"{k[0][:44]:44s} {k[1][:28]:28s} 1024: {min(a):8.4f}  3072: {min(b):8.4f}  {100*(min(b)/min(a)-1):+5.1f} %")
PY
tools/ab_cfg.sh dot "HE355_LIB_PATH=$A" "HE355_LIB_PATH=$B"
tools/ab_cfg.sh bfv_matmul "HE355_LIB_PATH=$A" "HE355_LIB_PATH=$B"
