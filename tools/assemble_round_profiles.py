#!/usr/bin/env python3
"""Puts the lane sweep / training-loop numbers of `tools/round_check_r4.sh learner <tag>` (gpurun_out/<tag>/) at the TOP of
profiles/r04_lanes_32768_anatomy.txt and profiles/r04_train_pipeline_ab.txt (replacing the top section of an earlier tag, keeping the history below
the ===== line) and copies the per-tag files (pipeline slowdown, collective) into profiles/.
    python tools/assemble_round_profiles.py r04_v56"""
import os, re, shutil, sys
tag = sys.argv[1]
T = os.path.join("gpurun_out", tag)
SEP = "=" * 150


def lines(f, pred):
    return [l.rstrip("\n") for l in open(os.path.join(T, f)) if pred(l)]


def history(path):
    s = open(path).read()
    return s[s.index(SEP):] if SEP in s else SEP + "\nEARLIER IN THE ROUND:\n\n" + s


p = "profiles/r04_lanes_32768_anatomy.txt"
new = [f"Round 4, build {tag} (HEAD: the expansion in one launch, rows in four arenas, node lists in four parts, the root pass on f16 hi + lo planes, the dense",
       "value-net modes in the fused boundary launch).  Same tool, same columns as below (tools/lanes_study.py); 'expand' = the doubles plies' group (now inside the expansion",
       "launch: ~0), 'leaves' = expand_all_kernel, 'apply' = the boundary launch (with the root pass from 24 576 lanes in f32).", ""]
new += [l[:330] for l in lines("lanes_study.txt", lambda l: "lanes" in l and "eager" in l)]
new += ["", "rocprofv3 --kernel-trace of 32 768 lanes, f32 (tools/lanes_study.py --only 32768,f32,nofork; --timeline):"]
new += lines("lanes_32768_timeline.txt", lambda l: True)
new += ["... and bf16 (config 5's self-play mode):"] + lines("lanes_32768_timeline_bf16.txt", lambda l: True) + [""]
open(p, "w").write("\n".join(new) + "\n" + history(p))

p = "profiles/r04_train_pipeline_ab.txt"
new = [f"Build {tag} (HEAD), same tool and definitions as below (tools/train_pipeline.py; the learner kernels are unchanged since r04_v48, the self-play step is faster):", ""]
new += [l[:300] for l in lines("train_pipeline.txt", lambda l: re.match(r"^(round|round_pipe|cont|cont_pipe) ", l) or " x the turns/s" in l)]
new += ["", "... with the delayed update (--delay 1):"] + [l[:300] for l in lines("train_pipeline_delay1.txt", lambda l: "lanes" in l)] + [""]
open(p, "w").write("\n".join(new) + "\n" + history(p))

shutil.copy(os.path.join(T, "pipeline_kernel_slowdown.txt"), f"profiles/{tag}_pipeline_kernel_slowdown.txt")
open(f"profiles/{tag}_train_dist_step.txt", "w").write(
    f"tools/train_dist_step.py 16384 256 1024 2048 on build {tag} (see r04_train_dist_step.txt for the columns):\n" + "\n".join(lines("train_dist_step.txt", lambda l: "slots," in l)) + "\n")
print("profiles/ updated for", tag)
