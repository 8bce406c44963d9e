"""TEST INFRASTRUCTURE (oracle/): the CPU oracle's TriForce runs on the small_gamma6 fixture for N injected-uniform
streams, cached as tests/golden/stochastic_oracle_runs.json so that the device-side paired test
(tests/test_gpu_e2e.py::test_stochastic_triforce_with_injected_uniforms) can compare >= 128 runs without spending
GPU-box time on the CPU side.  The oracle (oracle/ref_model.py) is pinned token-for-token to the unmodified reference on
this fixture (oracle/gen_golden.py, tests/test_oracle_golden.py); this script only replays it with other uniforms:
seed s uses tests.helpers.fixed_uniforms(n=2048, seed=500 + s), the stream the test injects on the device.

    python oracle/gen_stochastic_runs.py [runs=160]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_model as M  # noqa: E402
from tests import helpers as Hh  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    g = Hh.load_golden("small_gamma6")
    oeng, _, _ = Hh.build_oracle(g, temperature=0.6, top_p=0.9)
    prompt = Hh.prompt_of(g)
    out = {"fixture": "small_gamma6", "temperature": 0.6, "top_p": 0.9, "max_len": 24, "gamma": g["gamma"],
           "uniforms": "tests.helpers.fixed_uniforms(n=2048, seed=500 + run)", "runs": []}
    t0 = time.time()
    for s in range(runs):
        us = Hh.fixed_uniforms(n=2048, seed=500 + s)
        r = M.triforce(oeng, prompt, g["gamma"], 24, 0.6, 0.9, rng=M.InjectedRng(us))
        out["runs"].append({"seed": 500 + s, "tokens": [int(t) for t in r["tokens"]], "accepted": int(r["accepted"]),
                            "drafted": int(r["drafted"]), "n": int(r["n"]), "steps": len(r["counts"])})
        if s % 16 == 15:
            print(f"{s + 1} runs, {time.time() - t0:.0f} s", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "stochastic_oracle_runs.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
