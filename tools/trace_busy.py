"""GPU occupancy of a run from a rocprofv3 --kernel-trace CSV: how much of the wall time between the first and the last kernel of
the timed region has at least one kernel running (union of the kernel intervals), the mean number of kernels in flight, and the
idle gaps by length.  Usage: python tools/trace_busy.py <kernel_trace.csv> [last_ms]  (only the last `last_ms` milliseconds before the
last kernel are analysed -- the timed region of bench.py is at the end of its run; default 800)."""
import csv
import sys


def main():
    path = sys.argv[1]
    last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
    iv = []
    with open(path) as f:
        for r in csv.DictReader(f):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    iv.sort()
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    if len(sys.argv) > 3 and sys.argv[3] == "bins":      # busy fraction per 100 ms of the whole trace: where is the timed region?
        nb = int((t1 - t0) / 1e8) + 1
        busy = [0.0] * nb
        cnt = [0] * nb
        cs, ce = iv[0][0], iv[0][1]
        merged = []
        for s_, e_, _ in iv[1:]:
            if s_ > ce:
                merged.append((cs, ce))
                cs, ce = s_, e_
            else:
                ce = max(ce, e_)
        merged.append((cs, ce))
        for s_, e_ in merged:
            b0, b1 = int((s_ - t0) / 1e8), int((e_ - t0) / 1e8)
            for b in range(b0, b1 + 1):
                lo, hi = t0 + b * 1e8, t0 + (b + 1) * 1e8
                busy[b] += max(0.0, min(e_, hi) - max(s_, lo))
        for s_, _, _ in iv:
            cnt[int((s_ - t0) / 1e8)] += 1
        print("busy fraction per 100 ms bin (kernels started):")
        print(" ".join(f"{b / 1e8:.2f}({c})" for b, c in zip(busy, cnt)))
        return
    cut = t1 - last_ms * 1e6
    iv = [x for x in iv if x[0] >= cut]
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    busy, cur_s, cur_e, gaps, where = 0, iv[0][0], iv[0][1], [], []
    last_name = iv[0][2]
    for s, e, name in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            where.append((s - cur_e, (cur_e - t0) / 1e6, last_name[:60], name[:60]))
            cur_s, cur_e = s, e
        else:
            if e > cur_e:
                cur_e, last_name = e, name
        if s > cur_e - 1 and e >= cur_e:
            last_name = name
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _ in iv)
    wall = t1 - t0
    print(f"kernels {len(iv)}, wall {wall / 1e6:.1f} ms, at least one kernel running {busy / wall:.4f} of it, "
          f"mean kernels in flight {total / wall:.2f} (sum of durations {total / 1e6:.1f} ms)")
    gaps.sort(reverse=True)
    for lim in (1e3, 1e4, 1e5, 1e6):
        g = [x for x in gaps if x >= lim]
        print(f"  idle gaps >= {lim / 1e3:.0f} us: {len(g)}, {sum(g) / 1e6:.2f} ms")


    where.sort(reverse=True)
    for g, at, before, after in where[:25]:
        print(f"  gap {g / 1e3:8.1f} us at +{at:7.1f} ms  after {before}  before {after}")


if __name__ == "__main__":
    main()
