"""Build helper of the CPU-side model builds (tests/mock_hip): one compile at a time per output, written under a temporary name and
renamed into place -- pytest-xdist workers that find the same stale .so would otherwise compile it concurrently and one of them
dlopen()s a half-written file ("file too short")."""
import fcntl
import os
import subprocess


def build_if_stale(out, deps, cmd_without_output):
    """`cmd_without_output` + ["-o", out] runs when `out` is missing or older than any of `deps`; safe under concurrent callers."""
    def stale():
        return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)

    if not stale():
        return
    with open(out + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if stale():  # (another worker may have built it while this one waited)
                tmp = f"{out}.tmp{os.getpid()}"
                subprocess.check_call([*cmd_without_output, "-o", tmp])
                os.replace(tmp, out)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
