#!/usr/bin/env python3
"""Why fwd16_decode_block interleaves the heads of an XCD in pairs: list scheduling of a causal launch's workgroups (durations
= fixed cost + tiles traversed) on the 32 compute units of one XCD, dispatched in block-index order to the unit that frees up
first.  Durations from the measured one-block backward kernels (D = 128: 25 us fixed, 1.375 us per 64-key tile, 16 row blocks of
256 rows at N = 4096: block b traverses 4 (b + 1) tiles).  python tools/sim_dispatch_order.py"""
import heapq


def makespan(order, units=32):
    h = [0.0] * units
    heapq.heapify(h)
    for d in order:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)


def main():
    fixed, per_tile, nb = 25.0, 1.375, 16
    dur = lambda b: fixed + per_tile * 4 * (b + 1)   # noqa: E731
    print("# causal launch, N = 4096 (16 blocks per head, longest first), one XCD = 32 compute units; time in us")
    print("heads per XCD | one head after the other | pairs of heads interleaved | ideal (sum / 32)")
    for heads in (2, 4, 8, 16, 32):
        serial = [dur(b) for h in range(heads) for b in range(nb - 1, -1, -1)]
        paired = []
        for g in range(0, heads, 2):
            paired += [dur(b) for b in range(nb - 1, -1, -1) for h in range(min(2, heads - g))]
        print("%13d | %24.1f | %26.1f | %.1f" % (heads, makespan(serial), makespan(paired), sum(serial) / 32))


if __name__ == "__main__":
    main()
