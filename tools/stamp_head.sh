#!/bin/bash
# Run in the container before a gpurun profile call: the GPU box gets no .git, so the commit the profile is taken from
# travels as .git_head (git-ignored; tools/make_traffic.py reads it there and checks it against the sources' hash).
cd "$(dirname "$0")/.."
python3 - <<'PY'
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "genefuserust_amd/csrc", "include/gfmatch.h"],
                            capture_output=True, text=True).stdout.strip())
json.dump({"git_head": head, "git_dirty_csrc": dirty, "kernel_src_sha": bench.kernel_source_sha()}, open(".git_head", "w"))
print(open(".git_head").read())
PY
