"""Compile the C part of the oracle (test infrastructure) with gcc: oracle/glibc_log.c ->
oracle/_build/libglibc_log.so.  Nothing under autompc_amd/ uses it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libglibc_log.so")
SRC = os.path.join(HERE, "glibc_log.c")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # -ffp-contract=off: every fused multiply-add of the restatement is written out explicitly
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", OUT + ".tmp", SRC,
                    "-lm", "-ldl"], check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
