#!/usr/bin/env python3
"""Evaluate a trained retrieval model by following YAML scenarios -- the MI355X build
of ``mdir/examples/iccv19/eval.py``.

    ./eval.py <shortcut | a.yml b.yml ...>

Same surface as the reference: a single non-.yml argument NAME expands to
``eval.yml eval_NAME.yml`` (eval.py:33-35); scenarios are deep-overlaid left to right
(:39-42); the merged dict must have exactly ``network`` / ``validation`` / ``data``
(validate.py:22); the three headline scores are printed as ``round(100*value, 2)``
(:53-62).  Scenario files are looked up in the current directory first, then in
``scenarios/`` next to this script.  Nothing is downloaded (the reference calls
``download_test`` at import, eval.py:29): datasets live under ``$CIRTORCH_ROOT/data/test``.
"""
import os.path
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from mdir_amd.scenario import dict_deep_overlay   # noqa: E402
from mdir_amd import stages                        # noqa: E402

SCORES = {
    "roxford5k/validation/score:ap_medium_avg.4": "roxford.5k medium",
    "rparis6k/validation/score:ap_medium_avg.4": "rparis.6k medium",
    "247tokyo1k/validation/score:ap_avg.4": "247tokyo.1k",
}


def find_scenario(name):
    for cand in (name, os.path.join(HERE, "scenarios", name)):
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(name)


def load_scenarios(names):
    if len(names) == 1 and not names[0].endswith(".yml"):
        names = ["eval.yml", "eval_%s.yml" % names[0]]
    scenario = {}
    for name in names:
        with open(find_scenario(name), "r") as handle:
            scenario = dict_deep_overlay(scenario, yaml.safe_load(handle))
    return scenario


def init_distributed():
    """``python -m torch.distributed.run --nproc-per-node G eval.py ...``: one process per GPU; the
    retrieval score then shards extraction and the database over the ranks (mdir_amd/sharded.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from mdir_amd.sharded import private_miopen_caches
    private_miopen_caches(local)
    import datetime
    limit = datetime.timedelta(minutes=30)            # extraction of a rank's slice happens between collectives
    if os.environ.get("MDIR_AMD_DRYRUN_ONE_GPU") == "1":
        # functional dry run on a 1-GPU box: every rank on cuda:0, collectives staged through gloo
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", timeout=limit)
    else:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit)
    return dist.get_rank()


def main(argv):
    scenario = load_scenarios(argv)
    if not scenario:
        sys.stderr.write("Scenario needs to be specified\n")
        return 1
    rank = init_distributed()
    if rank != 0:
        import contextlib
        with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
            stages.validate(scenario, ())
        return 0
    metadata, = stages.validate(scenario, ())
    for heading, section in metadata.items():
        print("\n%s\n" % heading.capitalize())
        for key, value in section.items():
            if key in SCORES:
                print("    %-20s %s" % (SCORES[key], round(100 * value, 2)))
        print()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
