"""How `python bench.py --gpus N` becomes N ranks (one process per GPU).

The BASELINE metric is quoted "at 1/2/4/8 GPU".  A caller may start the ranks itself (`python -m torch.distributed.run ... bench.py --gpus N`:
RANK / LOCAL_RANK / WORLD_SIZE arrive in the environment) or run the plain command; in the second case the process re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port.  Either way the number of ranks that come up MUST equal --gpus: a mismatch is a hard
failure, never a silent 1-GPU measurement labelled otherwise.  No GPU or torch import happens here (the decision is taken before either)."""
import os
import socket
import sys


class LaunchError(SystemExit):
    """rank count and --gpus disagree"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch_command(script, script_args, gpus, env, executable=None, port=None):
    """None when this process is already what was asked for (one rank of `gpus`, or the only process of a 1-GPU run); otherwise the argv
    that starts `gpus` ranks of the same command.  Raises LaunchError when a launcher brought up a different number of ranks than --gpus."""
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != int(gpus):
            raise LaunchError("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to measure something else than what the line "
                              "would say (start `--nproc-per-node %d`, or run the plain command and let it launch its own ranks)" % (gpus, world, gpus))
        return None
    if int(gpus) <= 1:
        return None
    return [executable or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(script_args)


def check_world(gpus, world_env, dist_world):
    """after init_process_group: the environment's and the communicator's rank counts both equal --gpus"""
    if int(world_env) != int(gpus) or int(dist_world) != int(gpus):
        raise LaunchError("bench.py: --gpus %d, WORLD_SIZE %d, communicator world size %d: they must agree" % (gpus, world_env, dist_world))


def maybe_self_launch(script, script_args, gpus, env=None):
    """re-exec under torch.distributed.run when needed (does not return in that case: same PID, so a caller's timeout / signal still
    reaches the launcher, which forwards them to its ranks)"""
    env = os.environ if env is None else env
    cmd = self_launch_command(script, script_args, gpus, env)
    if cmd is None:
        return
    print("[bench] --gpus %d without a launcher: starting %d ranks: %s" % (gpus, gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execve(cmd[0], cmd, dict(env, LH_BENCH_SELF_LAUNCHED="1"))
