"""Hash of the kernel sources with comments and blank space removed: what identifies the BUILD that PMC counters were collected
on (profiles/pmc_traffic.json `_source_hash`, tools/make_pmc_traffic.py) to the bench that reports them (bench.py withholds
counters from any other build).  Editing a comment does not make the counters stale; editing code does."""
import hashlib
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussian-mesh-splatting_amd", "csrc")
_COMMENT = re.compile(rb"//[^\n]*|/\*.*?\*/", re.S)


def kernel_source_hash(csrc: str = _CSRC) -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                code = _COMMENT.sub(b"", f.read())
            h.update(name.encode() + b"\0" + b"\n".join(l.strip() for l in code.splitlines() if l.strip()))
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
