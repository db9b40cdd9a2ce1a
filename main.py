"""Launcher with the reference's CLI (code/main.py:9-31):

    python main.py --device 0,1,2,3 --config_file IDNet/sasrec.yaml overall/ID.yaml

One process per GPU via torch.distributed.run on a random master port; a single device runs run.py in-process
environment (no launcher).  OMP_NUM_THREADS=1 as in the reference (main.py:5); CUDA_LAUNCH_BLOCKING is NOT set --
the MI355X path is asynchronous end to end."""
import argparse
import os
import random
import subprocess
import sys

os.environ["TOKENIZERS_PARALLELISM"] = "false"
os.environ.setdefault("OMP_NUM_THREADS", "1")

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--device", default="0", type=str)
    parser.add_argument("--config_file", nargs="+")
    args = parser.parse_args()
    devices = args.device.split(",")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HIP_VISIBLE_DEVICES=args.device, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run_py = os.path.join(here, "run.py")
    if len(devices) == 1:
        cmd = [sys.executable, run_py, "--config_file", *args.config_file]
        env.update(LOCAL_RANK="0", RANK="0", WORLD_SIZE="1")
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(devices)}",
               "--master-addr", "127.0.0.1", "--master-port", str(random.randint(10002, 19999)), run_py,
               "--config_file", *args.config_file]
    sys.exit(subprocess.call(cmd, env=env))
