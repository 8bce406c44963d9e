"""Host pieces of the decode loop's inner hop (accept record seen -> next draft graph replay called -> call returned), from
inside the driver-form bench:  TRIFORCE_HOP_TRACE=1 python tools/hop_trace.py [bench flags]"""
import json
import os
import statistics
import sys

os.environ["TRIFORCE_HOP_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from triforce_amd.utils import decoding  # noqa: E402

if __name__ == "__main__":
    sys.argv = [sys.argv[0], "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--random-steps", "0"] + sys.argv[1:]
    try:
        bench.main()
    finally:
        tr = decoding._HOP_TRACE or []
        if tr:
            a = sorted(x[0] / 1e3 for x in tr)
            b = sorted(x[1] / 1e3 for x in tr)
            print(json.dumps({"hops": len(tr), "record_seen_to_replay_call_us": {"median": round(statistics.median(a), 1), "p10": round(a[len(a) // 10], 1), "p90": round(a[9 * len(a) // 10], 1)},
                              "inside_replay_call_us": {"median": round(statistics.median(b), 1), "p10": round(b[len(b) // 10], 1), "p90": round(b[9 * len(b) // 10], 1)}}))
