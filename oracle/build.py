"""Build recipe for the oracle shared library (gcc, a few seconds).  Outputs stay under oracle/."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgsplat_oracle.so")
SRC = os.path.join(HERE, "gsplat_oracle.c")


def build(force: bool = False) -> str:
    stale = (not os.path.exists(SO)) or os.path.getmtime(SO) < os.path.getmtime(SRC)
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-B", "libgsplat_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return SO


if __name__ == "__main__":
    print(build(force=True))
