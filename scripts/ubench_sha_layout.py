#!/usr/bin/env python3
"""Which part of the memory system caps the lane-per-stream SHA kernel? Same lane count and
bytes hashed per lane, different placement of the lanes' streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbs_plus_amd import Engine, buzhash

eng = Engine(buzhash.NewConfig(4 << 20))
total = 40 << 30
buf = eng.alloc(total)
eng.fill(buf.ptr, total, 1, 0)
SEG = 131136
def run(name, n, seg, stride):
    segs = [((i * stride) % (total - seg), seg) for i in range(n)]
    eng.sha256_many(buf, segs[:64], nbytes=total)
    t0 = time.perf_counter(); eng.sha256_many(buf, segs, nbytes=total); dt = time.perf_counter() - t0
    print(f"{name:34s} lanes={n:6d} seg={seg} stride={stride:>10d} ms={dt*1e3:8.2f} us/blk={dt*1e6/(seg//64+1):7.2f} GB/s={n*seg/dt/1e9:7.1f}", flush=True)
for n in (16384, 65536):
    run("same region (L2-hot, 1 page)", n, SEG, 0)
    run("packed 2 KiB apart (same pages)", n, SEG, 2048 + 64)          # overlapping streams, <=160 MiB span
    run("span 1 GiB (512 x 2MiB pages)", n, SEG, ((1 << 30) // n) + 64)
    run("span 8 GiB", n, SEG, ((8 << 30) // n) + 64)
    run("span 36 GiB", n, SEG, ((36 << 30) // n) + 64)
