"""Regenerates tests/golden/plasma_lens{,_short,_boosted}_3d_checksums.json from the reference checkout (read-only,
only in the build container): the golden checksums of Examples/Tests/plasma_lens/inputs_test_3d_plasma_lens,
_short and _boosted as the reference's own regression suite stores them."""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
for case in ("plasma_lens", "plasma_lens_short", "plasma_lens_boosted"):
    gold = json.load(open(os.path.join(ref, "Regression/Checksum/benchmarks_json/test_3d_%s.json" % case)))
    json.dump({"source": "Regression/Checksum/benchmarks_json/test_3d_%s.json of the reference "
                         "(Examples/Tests/plasma_lens/inputs_test_3d_%s, max_step 84); copied values, see "
                         "tests/golden/make_plasma_lens_golden.py" % (case, case),
               "rtol": 1e-9, "checksums": gold}, open(os.path.join(here, case + "_3d_checksums.json"), "w"), indent=1)
