# Round 5, call X (the round's last GPU seconds): the cfg 5 core step with the K-major GEMM forms on opaque LDS-DMA requests:
# FK_BWD_K_MAJOR=1 (default: weight gradients through layout 2) and =2 (data gradients through layout 1 too: no weight
# transposes) on the new build, =1 on the previous build (same box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05x_train_kmajor.txt
: > $O
one() { tag=$1; shift; env "$@" TRAIN_E2E=0 TRAIN_STEPS=3 timeout 120 python tools/train_prof.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag: cfg5 core step', round(d['ms_per_step'],1), 'ms  peak', round(d['peak_memory_gb'],1), 'GB  loss', d['loss'])" >> $O; }
one new_kmajor1 FK_BWD_K_MAJOR=1
one new_kmajor2 FK_BWD_K_MAJOR=2
one old_kmajor1 FK_BWD_K_MAJOR=1 FK_LIB_PATH=build_ab/before_kmajor/gpt_image_edit_amd/libfk_gfx950.so
cat $O
