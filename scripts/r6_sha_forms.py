"""Per-CU throughput of the SHA-256 kernel forms on full lanes: N equal segments hashed by pbsgpu_sha256_many_device with
sha_form 0 (wave pairs: sparse or dense by sha_dense_pct), 1 (single-wave lanes), 2 (express)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from pbs_plus_amd import Engine, buzhash
import hashlib

GiB = 1 << 30
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 10
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device('cuda:0')
data = torch.empty(seg * nseg // 8, dtype=torch.int64, device=dev).random_()
data = data.view(torch.uint8)
segs = [(i * seg, seg) for i in range(nseg)]
host = data[: seg].cpu().numpy().tobytes()
want0 = hashlib.sha256(host).digest()
for name, opts in (("pairs sparse", dict(sha_form=0, sha_dense_pct=1000000)), ("pairs dense", dict(sha_form=0, sha_dense_pct=1)),
                   ("lanes", dict(sha_form=1, sha_dense_pct=1000000)), ("lanes dense", dict(sha_form=1, sha_dense_pct=1)), ("express", dict(sha_form=2))):
    e = Engine(buzhash.NewConfig(4 << 20), **opts)
    out = e.sha256_many(data, segs)
    assert bytes(out[0]) == want0, name
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = e.sha256_many(data, segs); ts.append(time.perf_counter() - t0)
    t = min(ts)
    blocks = seg // 64 + 1
    print(f"{name:14s} {seg * nseg / GiB / t:8.1f} GiB/s  {seg*nseg/GiB/t/256:6.3f} GiB/s per CU   {t*1e3:8.1f} ms   chain-blocks per us and CU {nseg*blocks/t/1e6/256:6.1f}")
    e.close()
