#!/usr/bin/env python3
"""SHA-256 over the kernel sources (quantum_css_codes_amd/csrc/*.hip|*.h|*.cpp|Makefile and include/gf2hip.h), names and contents
in sorted order: what profiles/traffic.json was measured on.  bench.py recomputes it and says "traffic_stale": true when the sources
have changed since (there is no git on the GPU box; where there is, this equals `git diff --quiet <captured_at_commit> -- csrc`)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "quantum_css_codes_amd", "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".h", ".cpp")) or n == "Makefile")
    for path in [os.path.join(csrc, n) for n in names] + [os.path.join(ROOT, "include", "gf2hip.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(digest())
