#!/usr/bin/env python3
"""Stage the reference package for ONE `gpurun` call (VERDICT r04 "Next" #2; SURVEY.md section 8d last row).

`/root/reference` exists only in the build container.  `gpurun` ships the working tree (minus .git/, gpurun_out/ and
.gpurunignore), untracked files included, so a copy of `normflows/` in the git-ignored directory `.refstage/` travels to the GPU
box with the snapshot; there

    python tools/cpu_reference.py --ref .refstage --where "gpu box" --out gpurun_out/r05_cpu_reference_gpubox.json
    NF_REFERENCE_PATH=.refstage python -m pytest tests/test_gpu_parity.py -k reference_own_containers

time the reference's own CPU path on the box's host cores and run the real-container drop-in test.  The copy is NEVER added to git
(`.refstage/` is in .gitignore) and `--clean` removes it after the call; nothing under normalizing-flows_amd/ reads it.

    python tools/stage_reference.py            # copy /root/reference/normflows -> .refstage/normflows
    python tools/stage_reference.py --clean    # remove .refstage/
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, ".refstage")
SRC = os.environ.get("NF_REFERENCE_SRC", "/root/reference")


def main():
    if "--clean" in sys.argv:
        shutil.rmtree(STAGE, ignore_errors=True)
        print("removed", STAGE)
        return
    src = os.path.join(SRC, "normflows")
    if not os.path.isdir(src):
        sys.exit("no reference at %s" % src)
    shutil.rmtree(STAGE, ignore_errors=True)
    shutil.copytree(src, os.path.join(STAGE, "normflows"),
                    ignore=shutil.ignore_patterns("__pycache__", "*_test.py", "*.pyc"))
    n = sum(len(f) for _, _, f in os.walk(STAGE))
    print("staged %d files under %s (git-ignored; remove with --clean)" % (n, STAGE))


if __name__ == "__main__":
    main()
