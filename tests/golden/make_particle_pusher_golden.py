"""Regenerates tests/golden/particle_pusher_3d_checksums.json and radiation_reaction_3d_checksums.json from the reference checkout (read-only, only in the
build container): the golden checksums of Examples/Tests/particle_pusher/inputs_test_3d_particle_pusher as the
reference's own regression suite stores them."""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
gold = json.load(open(os.path.join(ref, "Regression/Checksum/benchmarks_json/test_3d_particle_pusher.json")))
json.dump({"source": "Regression/Checksum/benchmarks_json/test_3d_particle_pusher.json of the reference "
                     "(Examples/Tests/particle_pusher/inputs_test_3d_particle_pusher, max_step 10000); copied values, "
                     "see tests/golden/make_particle_pusher_golden.py",
           "rtol": 1e-9, "checksums": gold}, open(os.path.join(here, "particle_pusher_3d_checksums.json"), "w"), indent=1)
gold = json.load(open(os.path.join(ref, "Regression/Checksum/benchmarks_json/test_3d_radiation_reaction.json")))
json.dump({"source": "Regression/Checksum/benchmarks_json/test_3d_radiation_reaction.json of the reference "
                     "(Examples/Tests/radiation_reaction/inputs_test_3d_radiation_reaction, max_step 64); copied values, "
                     "see tests/golden/make_particle_pusher_golden.py",
           "rtol": 1e-9, "checksums": gold}, open(os.path.join(here, "radiation_reaction_3d_checksums.json"), "w"), indent=1)
