#!/bin/bash
# Round 6 (GPU box): the new boundary pieces on the device — skin modules, calibration, helper entry, unchanged reference drivers — and a headline line
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_check2
mkdir -p $O
timeout 1500 python -m pytest tests/test_skin_modules.py tests/test_calib.py tests/test_phymbl.py tests/test_reference_drivers.py tests/test_turb_series.py tests/test_abi.py tests/test_diagnostics.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -5 $O/tests.log
grep -E "^\{'|^\{\"" $O/tests.log | head -10
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r6_check2/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "value_norm", "ms_per_step", "calib")}, r["roofline"]["kernel_ms"], r["cpu_baseline"]["value"])
PY
timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 10 --verify > $O/bench8.json 2> $O/bench8.err
tail -c 600 $O/bench8.json
