import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
os.chdir("/root/repo")
import gpu_checks as G
for name, fn, kw in G.ALL_CHECKS:
    if name in ("train_direct", "train_gradcache", "train_7b_layer", "train_packed_vs_padded", "embed_scatter", "full_depth_parity_32_layers"):
        r = fn(**kw)
        print("RESULT", name, r["ok"], r["detail"], flush=True)
