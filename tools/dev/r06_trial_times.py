"""per-trial wall times by outcome (accepted / rejected / stay) of the headline bundle: PTAM_DEBUG_TRIAL_TIMES=1 in the library,
parsed here; `the trial after a rejected one` is what the second queue (PTAM_TWO_QUEUES=1: the first line) is about; the second line is the default.   usage: r06_trial_times.py [reps]"""
import os, sys, re, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", ".")
code = ("import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from ptam_cg_amd import host, synth\n"
        "from ptam_cg_amd._lib import load\n"
        "ctx = host.Context(lib=load())\n"
        "prob = synth.make_ba_problem(50, 5000, synth.SEED_BA if hasattr(synth, 'SEED_BA') else 11)\n"
        "for rep in range(%d):\n"
        "    ba = synth.load_into(host.Bundle(ctx, max_iterations=20, update_sq_conv_limit=0.0), prob)\n"
        "    ba.prepare(); ba.Compute(); ba.close()\n") % (R, int(sys.argv[1]) if len(sys.argv) > 1 else 12)
for one in ("", "1"):
    env = dict(os.environ, PTAM_DEBUG_TRIAL_TIMES="1")
    if one:
        env.pop("PTAM_TWO_QUEUES", None)
    else:
        env["PTAM_TWO_QUEUES"] = "1"   # (a rejected trial's continuation on the second queue; the decision is then a launch of its own)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    acc, rej, stay, after_rej, after_other = [], [], [], [], []
    for line in r.stderr.splitlines()[2:]:   # (the first calls: cold)
        if "trial times" not in line:
            continue
        toks = re.findall(r"([ars])([0-9.]+)", line.split(":", 1)[1])
        for i, (k, v) in enumerate(toks):
            if i == 0:
                continue
            v = float(v)
            {"a": acc, "r": rej, "s": stay}[k].append(v)
            (after_rej if toks[i - 1][0] == "r" else after_other).append(v)
    m = lambda x: sum(x) / len(x) if x else float("nan")
    print(f"one_queue={one or 0}: accepted {m(acc):.1f} us (n {len(acc)}) | rejected {m(rej):.1f} (n {len(rej)}) | stay {m(stay):.1f} (n {len(stay)}) | "
          f"a trial after a rejected one {m(after_rej):.1f} (n {len(after_rej)}) | after another {m(after_other):.1f}")
