#!/bin/bash
# round 5, job 1: GPU tests + LN-site parity + masked-observation A/B + MSEFast round-group A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 300 python tools/ln_site_parity.py > $O/ln_site_parity.txt 2>&1; tail -8 $O/ln_site_parity.txt
timeout 300 python tools/observe_alone_ab.py > $O/observe_alone_ab.txt 2>&1; cat $O/observe_alone_ab.txt | tail -40
for mib in 0 256 192 128 96 64; do
  OSQ_MSE_GROUP_MIB=$mib timeout 400 python bench.py --no-cpu-baseline --no-kernel-table --calib-configs 3 --steps 20 --warmup 5 > $O/mse_group_$mib.json 2> $O/mse_group_$mib.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mse_group_$mib.json").read().strip().splitlines()[-1])
    c = d["config"] if "calibration_config3" in d.get("config", {}) else d
    def find(o, k):
        if isinstance(o, dict):
            if k in o: return o[k]
            for v in o.values():
                r = find(v, k)
                if r is not None: return r
        return None
    print("group MiB $mib:", json.dumps(find(d, "calibration_config3"))[:600])
except Exception as e:
    print("group MiB $mib: failed", e)
PY
done
