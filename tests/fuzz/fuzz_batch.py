"""Fuzz of the asynchronous / batched entry points against the synchronous rr_simulate of the same context:
rr_simulate_device pipelined over the lanes, rr_simulate_batch_columns_device + rr_assemble_frames_device,
azimuth-sharded columns + rr_assemble_blocks_device, rr_simulate_material_sets_device -- random configs,
bit-exact images expected.  usage: fuzz_batch.py [iterations] [seed]"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from common import golden_beams, GOLDEN
sys.path.insert(0, GOLDEN)
import gen_oracle_images as gen


def run(iters=20, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    s = gen.two_room_scene()
    bad = 0
    dev = torch.device("cuda", 0)
    for it in range(iters):
        base_mats = params.kaist_materials() + [params.PENETRABLE]
        cfg = params.kaist_preset(
            n_reflections=int(rs.randint(1, 5)), ambient_noise=int(rs.choice([0, 1, 2])), n_cells=int(rs.choice([3424, 1000, 256])),
            signal_denoising=int(rs.choice([0, 1, 3])), scroll_image=int(rs.choice([0, 4, 100, 7, 399])),
            record_multi_path=bool(rs.randint(0, 2)), record_multi_reflection=bool(rs.randint(0, 2)))
        c = native.Context(0)
        c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder=str(rs.choice(["host", "gpu"])))
        c.set_materials(base_mats, s["object_materials"], 0); c.set_config(cfg); c.set_beam_samples(golden_beams(int(rs.randint(1, 60))))
        c.set_noise_offsets((rs.uniform(0, 1, 400) * 1000).astype(np.float32))
        K = int(rs.randint(1, 9))
        poses = np.stack([scenes.yaw_pose(float(rs.uniform(-2, 2)), float(rs.uniform(-2, 2)), float(rs.uniform(0, 1.5)),
                                          float(rs.uniform(-3, 3))) for _ in range(K)]).astype(np.float32)
        ref = np.stack([c.simulate(p)[0] for p in poses])
        st = torch.cuda.Stream(device=dev)
        sp = st.cuda_stream
        C = cfg.n_cells
        what = []
        # (a) pipelined rr_simulate_device, one buffer per frame
        imgs = torch.zeros((K, C, 400), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
        for k in range(K): c.simulate_device(poses[k], imgs[k].data_ptr(), sp)
        c.synchronize(sp); torch.cuda.synchronize()
        if not np.array_equal(imgs.cpu().numpy(), ref): what.append("simulate_device")
        # (b) frame batch + assemble_frames
        cols = torch.zeros((K, 400, C), dtype=torch.uint8, device=dev); imgs.zero_(); torch.cuda.synchronize()
        c.simulate_batch_columns_device(poses, 0, 400, cols.data_ptr(), sp)
        c.assemble_frames_device(cols.data_ptr(), 400, 400 * C, K, 400 * C, imgs.data_ptr(), sp)
        c.synchronize(sp); torch.cuda.synchronize()
        if not np.array_equal(imgs.cpu().numpy(), ref): what.append("batch_columns")
        # (b2) whole frames in one call
        imgs.zero_(); torch.cuda.synchronize()
        c.simulate_batch_device(poses, imgs.data_ptr(), sp)
        c.synchronize(sp); torch.cuda.synchronize()
        if not np.array_equal(imgs.cpu().numpy(), ref): what.append("batch_device")
        # (b3) host delivery: 5..9 batches of the K poses (rotated) on two streams, so that lanes and their copy records
        # are reused; interleaved at random with a device-path batch on the same lanes (which must flush what waits
        # there); waits in random order.  RR_FOLD_MIN_BUSY=0 (set by the caller's environment) forces the folded route
        NB = int(rs.randint(5, 10))
        st2 = torch.cuda.Stream(device=dev)
        hosts = [native.HostImages((K, C, 400)) for _ in range(NB)]
        for h in hosts: h.array[:] = 9
        for b in range(NB):
            ps = np.roll(poses, -b, axis=0)
            c.simulate_batch_host_async(ps, hosts[b].ptr, (sp, st2.cuda_stream)[b % 2])
            if rs.randint(0, 4) == 0:
                c.simulate_batch_device(poses, imgs.data_ptr(), sp)
        for b in rs.permutation(NB):
            c.wait_host(hosts[int(b)].ptr)
            if not np.array_equal(hosts[int(b)].array, np.roll(ref, -int(b), axis=0)): what.append("host_async[%d of %d]" % (int(b), NB)); break
        c.synchronize(sp); torch.cuda.synchronize()
        for h in hosts: h.close()
        # (c) sharded: W ranks' blocks of one frame, assembled from [rank][n_loc][C]
        W = int(rs.choice([2, 4, 8])); nl = 400 // W
        blocks = torch.zeros((W, nl, C), dtype=torch.uint8, device=dev); one = torch.zeros((C, 400), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for r in rs.permutation(W): c.simulate_columns_device(poses[0], int(r) * nl, (int(r) + 1) * nl, blocks[int(r)].data_ptr(), None, sp)
        c.assemble_blocks_device(blocks.data_ptr(), nl, nl * C, one.data_ptr(), sp)
        c.synchronize(sp); torch.cuda.synchronize()
        if not np.array_equal(one.cpu().numpy(), ref[0]): what.append("sharded")
        # (d) material sets vs one by one
        M = int(rs.randint(1, 6))
        sets = np.repeat(np.asarray([m.astuple() for m in base_mats], np.float32)[None], M, axis=0)
        sets[:, 1:, 0] = rs.choice([0.0, 0.05, 0.1, 0.2], (M, 2)); sets[:, 1:, 1:3] = rs.uniform(0, 1, (M, 2, 2)); sets[:, 1:, 3] = rs.choice([1.0, 5.0, 30.0, 3000.0], (M, 2))
        got = c.simulate_material_sets(poses[0], sets)
        for m in range(M):
            c.set_materials([params.RadarMaterial(*[float(x) for x in sets[m, i]]) for i in range(3)], s["object_materials"], 0)
            if not np.array_equal(c.simulate(poses[0])[0], got[m]): what.append("material_sets[%d]" % m); break
        c.close()
        if what:
            bad += 1
            print("MISMATCH iteration", it, what, cfg)
    if verbose:
        print("batch fuzz: %d iterations, %d mismatching" % (iters, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
