"""A fingerprint of the library's sources (kernels, host code, ABI headers): profiles/ records it beside what was measured,
bench.py recomputes it at run time — a committed profile summary is only quoted when it belongs to the code that is running
(the GPU box holds no git history: a commit id is not available there)."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc", ".h")) or f == "Makefile"]
    files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]   # (what the .so is built from: not the header-only C++ shim)
    for p in sorted(files):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
