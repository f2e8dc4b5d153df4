"""Timing ablations of the segmented forward kernel's bookkeeping (tuning build): python tools/ablate_gru_seg.py"""
import sys, os, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
if len(sys.argv) > 1:
    import torch
    from mm_dfn_amd import _hip
    _hip.set_tuning(True)
    H, dev, T, rows = 100, "cuda", 110, 80
    lib = _hip.lib()
    gi = torch.randn(T, rows, 600, device=dev); whh = [torch.randn(300, 100, device=dev) * 0.1 for _ in range(2)]
    bhh = [torch.randn(300, device=dev) * 0.1 for _ in range(2)]
    y = torch.zeros(T, rows, 200, device=dev); g = torch.zeros(T, rows, 2, 4, 100, device=dev)
    pa = _hip.ptr_array
    def f():
        lib.mmdfn_gru_seq_fwd_seg(1, pa([gi]), pa(whh), pa(bhh), pa([y]), pa([g]), _hip.int_array([rows]), _hip.int_array([T]), H,
                                  pa([None]), _hip.int_array([1]), _hip.int_array([1]), _hip.int_array([-1]), pa([None]), _hip.stream())
    def p():
        lib.mmdfn_gru_seq_fwd(1, pa([gi]), pa(whh), pa(bhh), pa([y]), pa([g]), _hip.int_array([rows]), _hip.int_array([T]), H, _hip.stream())
    for name, fn in (("plain", p), ("seg", f)):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        print("  %s %.1f us" % (name, e0.elapsed_time(e1) / 30 * 1e3))
else:
    for abl in ("0", "32", "64", "96", "128", "224"):
        print("MMDFN_GRU_ABL=%s  (32 no segment-start branch, 64 no schedule prefetch in the recurrence waves, 128 none in the I/O wave)" % abl)
        env = dict(os.environ, MMDFN_GRU_ABL=abl)
        if abl == "0":
            env.pop("MMDFN_GRU_ABL")
        subprocess.run([sys.executable, __file__, "run"], env=env)
