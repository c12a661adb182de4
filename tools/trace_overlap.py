"""Concurrency of a rocprofv3 --kernel-trace: sum of kernel durations / length of the union of their intervals, per stream count."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]) for r in rows)
tot = sum(e - s for s, e, _, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print("kernels %d  sum %.1f ms  union %.1f ms  span %.1f ms  overlap factor %.2f  queues %s" % (
    len(iv), tot / 1e6, union / 1e6, span / 1e6, tot / union, collections.Counter(q for _, _, q, _ in iv).most_common(10)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, _, k in iv:
    by[k][0] += 1; by[k][1] += e - s
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-42s n=%6d  total %.2f ms  avg %.2f us" % (k, n, t / 1e6, t / n / 1e3))
