"""sha256 over the sources that define the GPU code (ezrt_amd/csrc/hip/*, include/*): profiles under
profiles/ carry this stamp, and bench.py refuses to quote counters from a profile of other code."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_source_hash():
    h = hashlib.sha256()
    for d in ("ezrt_amd/csrc/hip", "include"):
        full = os.path.join(ROOT, d)
        for fn in sorted(os.listdir(full)):
            p = os.path.join(full, fn)
            if os.path.isfile(p):
                h.update(fn.encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]
