#!/bin/bash
# child order for visibility walks: (A) triangles, then nodes by descending area, slots taken in ascending order (the product);
# (B) the same tree walked from the highest slot down = smallest node first; (C) nodes in ascending area walked from the
# highest slot down = largest node first, triangles last
cd ${GRAFT_REPO_ROOT:-/root/repo}
visits() { python - <<'PY'
import sys
sys.path.insert(0, '.')
import rust_renderer_amd as rr
W, H = 1920, 1080
for cfg in (1, 3):
    scene = rr.scenes.scene_for_config(cfg, tex_size=64)
    r = rr.Renderer(W, H)
    scene.upload(r)
    r.set_option("count_visits", 1)
    loop = rr.FrameLoop(r, scene.make_view(W, H, use_ris_light_sampling=0))
    loop.frame(rr.PASS_REFERENCE_PT)
    c = r.get_stats()
    cl = c.rays[0] + c.rays[1]; sh = c.rays[2] + c.rays[3]
    print("   config %d closest nodes/ray %.2f tris/ray %.2f | shadow %.2f / %.2f" % (cfg, c.nodes_visited / cl, c.tris_tested / cl, c.shadow_nodes_visited / max(sh, 1), c.shadow_tris_tested / max(sh, 1)), flush=True)
PY
}
bench() { timeout -k 10 200 python bench.py --warmup 16 --no-cpu-baseline --no-alone "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.1f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
echo "A: descending area, ascending slots"; export UTOPIAN_HIP_LIB=$PWD/rust-renderer_amd/libutopian_hip.so; unset UH_CHILD_ASC; visits 2>&1 | grep config; bench --steps 64; bench --steps 64; bench --config 2 --steps 32
echo "B: descending area, descending slots"; export UTOPIAN_HIP_LIB=$PWD/rust-renderer_amd/libutopian_hip_rev.so; visits 2>&1 | grep config; bench --steps 64; bench --steps 64; bench --config 2 --steps 32
echo "C: ascending area, descending slots"; export UH_CHILD_ASC=1; visits 2>&1 | grep config; bench --steps 64; bench --steps 64; bench --config 2 --steps 32
