#!/usr/bin/env python3
"""Adds the MI355X backend to a checkout of uos/radarays_ros.

    python integration/apply.py <path to the radarays_ros checkout> [--dry-run]

1. copies integration/include/radarays_ros/RadarHIP.hpp and integration/src/radarays_ros/RadarHIP.cpp into the tree;
2. applies integration/patches/*.json: pure INSERTIONS, anchored by the line number of the unpatched file and guarded by
   the file's sha256 (the checkout this repository was written against: `main`, 2025-02-17) -- no text of the reference is
   stored here, so a file that differs is refused instead of being patched somewhere else.

Then, in the catkin workspace:  catkin build radarays_ros -DRADARAYS_MI355_DIR=<this repository>
and start the node with `_hip:=true` (optionally `_hip_devices:=[0,1,...]`, `_hip_build_on_gpu:=true`).
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
NEW_FILES = ["include/radarays_ros/RadarHIP.hpp", "src/radarays_ros/RadarHIP.cpp"]


def patched_text(original: str, patch: dict) -> str:
    """`original` with the patch's insertions; raises ValueError when the file is not the one the patch was made for."""
    digest = hashlib.sha256(original.encode()).hexdigest()
    if digest != patch["sha256"]:
        raise ValueError("%s: sha256 %s, the patch was made for %s" % (patch["file"], digest[:16], patch["sha256"][:16]))
    lines = original.split("\n")
    out, last = [], -1
    by_line = {}
    for ins in patch["insertions"]:
        if not (0 <= ins["after_line"] <= patch["n_lines"]) or ins["after_line"] <= last:
            raise ValueError("%s: insertions must be ascending and inside the file" % patch["file"])
        last = ins["after_line"]
        by_line[ins["after_line"]] = ins["text"]
    out.extend(by_line.get(0, []))
    for no, line in enumerate(lines, start=1):
        out.append(line)
        if no in by_line:
            out.extend(by_line[no])
    return "\n".join(out)


def load_patches():
    d = os.path.join(HERE, "patches")
    return [json.load(open(os.path.join(d, f))) for f in sorted(os.listdir(d)) if f.endswith(".json")]


def apply(tree: str, dry_run: bool = False) -> list:
    done = []
    texts = {}
    for patch in load_patches():        # check everything before anything is written
        path = os.path.join(tree, patch["file"])
        texts[path] = patched_text(open(path, newline="").read(), patch)
    for path, text in texts.items():
        if not dry_run:
            with open(path, "w", newline="") as f:
                f.write(text)
        done.append(path)
    for rel in NEW_FILES:
        dst = os.path.join(tree, rel)
        if not dry_run:
            shutil.copyfile(os.path.join(HERE, rel), dst)
        done.append(dst)
    return done


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 1:
        sys.exit(__doc__)
    try:
        for p in apply(args[0], "--dry-run" in sys.argv):
            print("ok ", p)
    except (ValueError, OSError) as e:
        sys.exit("refused: %s" % e)
