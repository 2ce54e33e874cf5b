import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, xxhash
from pbs_plus_amd import Engine, buzhash
eng = Engine(buzhash.NewConfig(4096))
rng = np.random.default_rng(1)
blob = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
case = sys.argv[1]
if case == "short":
    segs = [(i * 300, i) for i in range(0, 241)]
elif case == "long_aligned":
    segs = [(i * 8192, 1024 + 64 * i) for i in range(40)]
elif case == "long_unaligned":
    segs = [(i * 8192 + 1 + (i % 7), 1000 + 61 * i) for i in range(40)]
elif case == "one_long":
    segs = [(0, 5000)]
elif case == "mixed":
    segs = [(i * 8192 + (i % 5), (i * 97) % 3000) for i in range(64)]
got = eng.xxh3_many(blob, segs)
bad = [(n, hex(int(g))) for (o, n), g in zip(segs, got) if int(g) != xxhash.xxh3_64_intdigest(blob[o:o + n].tobytes())]
print(case, "bad:", bad[:5], "of", len(segs))
