"""which sharded-plan configuration reports an overflow / wrong rows (debug aid for tests/test_gpu_dist_plan.py)"""
import os, sys
sys.path[:0] = ["tests", "."]
import torch
import test_gpu_dist_plan as T

cases = [(8, False, [3, 4]), (8, False, [3, 20]), (8, False, [3, 40]), (8, False, [6, 60]), (8, False, [12, 30]), (4, False, [3, 60]),
         (5, False, [3, 60]), (8, False, [3, 60])]
if len(sys.argv) > 1:
    cases = eval(sys.argv[1])
for world, project, fan in cases:
    T.FAN = fan
    try:
        T.test_all_ranks_in_one_process_end_to_end(world, project, torch.float32)
        print(world, project, fan, "ok", flush=True)
    except AssertionError as e:
        print(world, project, fan, "FAILED", str(e)[:160].replace("\n", " "), flush=True)
