import sys, ctypes as C
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0)
lib = L.lib(); lib.pg_debug_read_probe.restype = C.c_int32
for bytes_ in (800_000_000, 2_000_000_000):
    for eb in (8, 4):
        for nt in (0, 1):
            for blocks in (1280, 2048, 4096, 8192):
                g = C.c_double()
                L.check(lib.pg_debug_read_probe(C.c_int64(bytes_), eb, nt, blocks, 10, C.byref(g)))
                print(f"bytes={bytes_/1e6:.0f}MB elem={eb}B nt={nt} blocks={blocks}: {g.value:.0f} GB/s", flush=True)
