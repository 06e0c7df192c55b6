"""The HBM-bound sub-operations of the iteration in isolation (txt2vid_amd.util.roofline.hbm_bound_lines): prints one JSON
object. Also the workload of the PMC passes behind profiles/r03_pmc_hbm.json:
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_hbm_fetch -- python3 $R/tools/hbm_micro.py 3
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_hbm_write -- python3 $R/tools/hbm_micro.py 3
    python3 tools/pmc_hbm.py $O/pmc_hbm_fetch $O/pmc_hbm_write > profiles/r03_pmc_hbm.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from txt2vid_amd.util.roofline import hbm_bound_lines          # noqa: E402

print(json.dumps(hbm_bound_lines(iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20)))
