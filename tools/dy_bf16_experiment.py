"""GPU: what storing the inter-layer gradient as bf16 would cost in ACCURACY, on the real kernels (csrc/train.cpp SNERF_TRAIN_DY_BF16: dL/dY and
dL/dZ of every activation-backward-fused layer rounded to bf16 in place).  Runs the reference-pinned training tests with and without the switch and prints what they
print: the worst relative gradient error against the reference's own gradients (budget 5e-4, measured 1-3e-4) and whether the reference's 24-step trajectory at the
benchmark's width is still followed within its band."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for arm, env in (("fp32 dY (shipped)", {}), ("bf16 dY (experiment)", {"SNERF_TRAIN_DY_BF16": "1"})):
    print(f"== {arm}", flush=True)
    for sel in ("train_step_vs_reference and W256", "full_size_training_step", "follows_the_reference_trajectory and W256 and False"):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_train.py", "-q", "-m", "gpu", "-s", "-k", sel], cwd=REPO, env=dict(os.environ, **env),
                           capture_output=True, text=True)
        lines = [l.strip() for l in r.stdout.splitlines() if re.search(r"worst relative gradient|passed|failed|step +\d+|max |Error|assert ", l)]
        print(f"  -k '{sel}'")
        for l in lines[-8:]:
            print("     ", l[:220], flush=True)
