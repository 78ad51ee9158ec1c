"""Regenerates tests/golden/pec_field_3d_checksums.json from the reference checkout
(read-only, only available in the build container): the golden checksums of
Examples/Tests/pec/inputs_test_3d_pec_field as the reference's own regression suite stores them."""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
gold = json.load(open(os.path.join(ref, "Regression/Checksum/benchmarks_json/test_3d_pec_particle.json")))
json.dump({"source": "Regression/Checksum/benchmarks_json/test_3d_pec_particle.json of the reference "
                     "(Examples/Tests/pec/inputs_test_3d_pec_particle, max_step 20); copied values, see "
                     "tests/golden/make_pec_field_golden.py",
           "rtol": 1e-9, "checksums": gold}, open(os.path.join(here, "pec_particle_3d_checksums.json"), "w"), indent=1)
gold = json.load(open(os.path.join(ref, "Regression/Checksum/benchmarks_json/test_3d_pec_field.json")))
out = {"source": "Regression/Checksum/benchmarks_json/test_3d_pec_field.json of the reference "
                 "(Examples/Tests/pec/inputs_test_3d_pec_field, max_step 125); copied values, see "
                 "tests/golden/make_pec_field_golden.py",
       "rtol": 1e-9, "checksums": gold}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pec_field_3d_checksums.json"), "w"), indent=1)
