import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "web-rwkv-gguf_amd"))
import microbench as mb, wrk
ctx = wrk.Context(0)
for n in (2, 4, 8, 16):
    mb.run(ctx, "Q4_K", 8192, 2048, nin=n, reps=50, turbo=True)
mb.run(ctx, "Q5_K", 14336, 4096, nin=16, reps=50, turbo=True)
mb.run(ctx, "Q4_K", 2048, 8192, nin=16, reps=50, turbo=True)
ctx.close()
