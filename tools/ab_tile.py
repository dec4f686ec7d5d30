import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tools")); sys.path.insert(0, os.path.join(R, "web-rwkv-gguf_amd"))
import microbench as mb, wrk
ctx = wrk.Context(0)
for (k, m, n) in ((2048, 8192, 512), (2048, 2048, 4096), (2048, 8192, 4096), (8192, 2048, 4096)):
    mb.run(ctx, "Q4_K", k, m, nin=n, reps=30, turbo=True)
ctx.close()
