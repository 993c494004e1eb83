#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: start [us since the step's first kernel], duration, stream, grid, kernel.

    python3 tools/b1_timeline.py <kernel_trace.csv> [first kernel of a step = embed_kernel] [step, counted from the end = 1] [steps shown = 1]

Takes the LAST complete step in the trace (from one `embed_kernel` to the next), or the n-th from the end (the streaming vocoder keeps two
chunks in flight: its last step is the drained pipeline, a step from the middle shows the overlap).  A tuning aid for the B = 1 latency path, where the order
and overlap of ~190 short launches matter more than any one kernel's rate."""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    first = sys.argv[2] if len(sys.argv) > 2 else "embed_kernel"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    def short(n):
        n = re.sub(r"^(void )?e2etts::\(anonymous namespace\)::", "", n)
        return re.sub(r"\(.*$", "", n)
    starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == first]
    if len(starts) < 2:
        raise SystemExit("need two steps in the trace")
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    span = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # the streaming vocoder alternates two streams: two markers = one step of each
    if len(starts) < back + span:
        raise SystemExit(f"the trace has {len(starts)} steps, {span} steps ending {back} from the end do not exist")
    a, b = starts[-back - span], starts[-back]
    t0 = int(rows[a]["Start_Timestamp"])
    streams = {}
    print("# start_us  dur_us  stream  threads  kernel")
    for r in rows[a:b]:
        s = streams.setdefault(r.get("Stream_Id", r.get("Queue_Id", "0")), len(streams) + 1)
        name = short(r["Kernel_Name"])
        threads = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  s{s}  {threads:8d}  {name}")
    print(f"# {'step' if span == 1 else f'{span} steps'}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us between first kernels")


if __name__ == "__main__":
    main()
