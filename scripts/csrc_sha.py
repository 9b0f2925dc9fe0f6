"""sha256 (first 16 hex digits) over the library's sources (holoagent_amd/csrc/*.hip, *.h, *.inl + include/*.h, sorted by name): the
tag a PMC pass's summary carries, so that bench.py can say whether the counters it quotes were taken on the build it is timing."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "holoagent_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "holoagent_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "holoagent_amd", "csrc", "*.inl")) + glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
