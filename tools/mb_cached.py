import sys, os
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0,'/root/repo')
import microbench as mb, wrk
ctx = wrk.Context(0)
for copies in (1, 4):
    print("copies", copies)
    for kind,k,m in [("Q4_K",2048,2048),("Q4_K",2048,8192),("Q4_K",8192,2048),("Q6_K",2048,65536),("F16",2048,4096)]:
        mb.run(ctx, kind, k, m, copies=copies)
ctx.close()
