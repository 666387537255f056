#!/bin/bash
# round-2 regression gate for a grower change (GPU box): parity suite, repeated fuzz, repeated full-size growth, timing
export BS_CLOUD_CACHE=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
python tests/tools/fuzz_parity.py --cases 300 --seed 4242 --repeat 3 --audit 2>&1 | tail -3
python tests/tools/check_large.py facade_1m 8 2>&1 | tail -3
bash tools/ab_dbg.sh facade_1m buildingsegment_amd/libbuildingsegment_hip.so
bash tools/ab_dbg.sh urban_10m buildingsegment_amd/libbuildingsegment_hip.so
bash tools/ab_dbg.sh urban_50m buildingsegment_amd/libbuildingsegment_hip.so
