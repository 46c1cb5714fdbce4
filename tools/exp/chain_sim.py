"""Discrete-event model of conv_chain_kernel's schedule: items (conv, tile, group) dealt round-robin (or dynamically) to G workgroups; an item of conv c
starts when the workgroup is free AND the 3x3 tile neighbourhood of conv c-1 is complete.  Cost = stages * t_stage + t_item.  Prints time per dense block.
usage: python tools/exp/chain_sim.py B H nrdb [G] [dynamic]"""
import sys, heapq
B, H, NR = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
G = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dyn = len(sys.argv) > 5
import os
TH = int(os.environ.get("TH", "32"))
T_STAGE, T_ITEM = float(os.environ.get("T_STAGE", 3.0 * TH / 32)), float(os.environ.get("T_ITEM", "3.0"))
tx, ty = (H + 31) // 32, (H + TH - 1) // TH
convs = [(8, 1), (12, 1), (16, 1), (20, 1), (24, 2)] * NR          # (stages, groups)
items = []
for c, (st, gr) in enumerate(convs):
    for b in range(B):
        for y in range(ty):
            for x in range(tx):
                for g in range(gr):
                    items.append((c, b, y, x, st))
done_t = {}                                                          # (c, b, y, x) -> completion time of all its groups
cnt = {}
def ready_time(c, b, y, x):
    if c == 0: return 0.0
    t = 0.0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            yy, xx = y + dy, x + dx
            if 0 <= yy < ty and 0 <= xx < tx:
                k = (c - 1, b, yy, xx)
                if k not in done_t: return None
                t = max(t, done_t[k])
    return t
free = [0.0] * G
if not dyn:
    # static: workgroup w takes items w, w+G, ...; process in global order (a dependency always points to an earlier item)
    for i, (c, b, y, x, st) in enumerate(items):
        w = i % G
        r = ready_time(c, b, y, x)
        assert r is not None
        s = max(free[w], r)
        e = s + st * T_STAGE + T_ITEM
        free[w] = e
        k = (c, b, y, x)
        cnt[k] = cnt.get(k, 0) + 1
        if cnt[k] == convs[c][1]: done_t[k] = max(e, done_t.get(("p", k), 0.0))
        else: done_t[("p", k)] = e
else:
    h = [(0.0, w) for w in range(G)]
    heapq.heapify(h)
    for i, (c, b, y, x, st) in enumerate(items):
        t, w = heapq.heappop(h)
        r = ready_time(c, b, y, x)
        s = max(t, r)
        e = s + st * T_STAGE + T_ITEM
        heapq.heappush(h, (e, w))
        free[w] = e
        k = (c, b, y, x)
        cnt[k] = cnt.get(k, 0) + 1
        if cnt[k] == convs[c][1]: done_t[k] = max(e, done_t.get(("p", k), 0.0))
        else: done_t[("p", k)] = e
total = max(free)
work = sum(st * T_STAGE + T_ITEM for (_, _, _, _, st) in items) / G
print("B=%d H=%d, %d dense blocks, G=%d, %s: %.1f us per block (perfect balance %.1f; launch-per-conv model %.1f)" % (
    B, H, NR, G, "dynamic" if dyn else "static", total / NR, work / NR,
    sum(-(-B * tx * ty * gr // G) * (st * T_STAGE + T_ITEM) for st, gr in convs[:5])))
