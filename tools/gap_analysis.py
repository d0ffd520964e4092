"""Idle-time analysis of a rocprofv3 kernel trace of `bench.py`: how much of a denoising step's wall time has NO kernel running on the GPU,
and how the gaps between consecutive dispatches are distributed.

    rocprofv3 --kernel-trace -d gpurun_out/gap -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-train-forward
    python tools/gap_analysis.py gpurun_out/gap [--steps 8]

Takes the LAST `steps` steps' worth of dispatches: the window is found from the per-step marker kernel `cfg_dpm_step_kernel` (one launch per
step).  Prints the union-busy time, the idle time, and the ten kernels whose completion is most often followed by idle time."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 8
    files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        sys.exit("no *kernel_trace.csv under " + root)
    rows = list(csv.DictReader(open(files[-1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    marks = [i for i, e in enumerate(ev) if "cfg_dpm_step" in e[2]]
    if len(marks) < steps + 1:
        sys.exit(f"only {len(marks)} step markers in the trace")
    lo, hi = marks[-steps - 1] + 1, marks[-1] + 1
    win = ev[lo:hi]
    t0, t1 = win[0][0], max(e[1] for e in win)
    busy, cur_end, idle_after = 0, t0, defaultdict(lambda: [0, 0])
    last_name = None
    gaps = []
    for s, e, name in win:
        if s > cur_end:
            gaps.append(s - cur_end)
            if last_name is not None:
                idle_after[last_name][0] += s - cur_end
                idle_after[last_name][1] += 1
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end, last_name = e, name
    wall = t1 - t0
    print(f"window: {steps} steps, {len(win)} dispatches ({len(win) / steps:.0f} per step), wall {wall / 1e6 / steps:.3f} ms per step")
    print(f"  some kernel running : {busy / 1e6 / steps:.3f} ms per step ({100.0 * busy / wall:.1f} %)")
    print(f"  GPU idle            : {(wall - busy) / 1e6 / steps:.3f} ms per step ({100.0 * (wall - busy) / wall:.1f} %) in {len(gaps) / steps:.0f} gaps per step")
    ksum = sum(e - s for s, e, _ in win)
    print(f"  sum of kernel durations: {ksum / 1e6 / steps:.3f} ms per step (overlap factor {ksum / busy:.2f})")
    if gaps:
        gaps.sort()
        q = lambda f: gaps[min(len(gaps) - 1, int(f * len(gaps)))] / 1e3
        print(f"  gap us: median {q(0.5):.1f}, p90 {q(0.9):.1f}, p99 {q(0.99):.1f}, max {gaps[-1] / 1e3:.1f}")
    print("  idle time by the kernel that ended before the gap (ms per step, gaps per step, name):")
    for name, (ns, n) in sorted(idle_after.items(), key=lambda kv: -kv[1][0])[:10]:
        print(f"    {ns / 1e6 / steps:7.3f}  {n / steps:6.1f}  {name[:110]}")


if __name__ == "__main__":
    main()
