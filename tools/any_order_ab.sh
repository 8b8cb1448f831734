cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in libutopian_hip_prev.so libutopian_hip.so; do
UTOPIAN_HIP_LIB=$PWD/rust-renderer_amd/$lib python - <<'PY'
import sys, time
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
    print("config %d closest nodes/ray %.2f tris/ray %.2f | shadow %.2f / %.2f" % (cfg, c.nodes_visited / cl, c.tris_tested / cl, c.shadow_nodes_visited / max(sh, 1), c.shadow_tris_tested / max(sh, 1)), flush=True)
PY
done
