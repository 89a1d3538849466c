"""build with -DPBWTAMD_WALKSTAT (hipcc ... -DPBWTAMD_WALKSTAT -o pbwt_amd/libpbwtgpu_walkstat.so; PBWTAMD_LIB=that): how long are the scans of the -stats sweep that leave a wave's own 256 positions?  python tools/walkstat.py [M] [sites] [kind]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd as amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = amd.load_library()
eng = amd.Engine(M, batch_sites=512)
panel = torch.empty((sites, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(panel.data_ptr(), 0, sites, seed=0x1A2B3C, kind=kind); eng.sync()
st = (ctypes.c_ulonglong * 16)()
L.pbwtamd_measure_walkstat(st, 1)
opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
eng.pass_begin(sites); eng.pass_advance(panel.data_ptr(), sites, sites, opts); eng.pass_end(opts); eng.sync()
L.pbwtamd_measure_walkstat(st, 0)
v = list(st)
print("M %d, %d sites, kind %d: per site: pending up %.0f down %.0f; LDS walks %.0f with %.0f steps (%.1f per walk); memory walks %.0f with %.0f steps (%.1f per walk)" % (
    M, sites, kind, v[12] / sites, v[13] / sites, v[0] / sites, v[1] / sites, v[1] / max(v[0], 1), v[2] / sites, v[3] / sites, v[3] / max(v[2], 1)))
print("   LDS walks by steps (1, 2, 3-4, 5-8, 9-16, 17-32, >32), per site:", " ".join("%.1f" % (x / sites) for x in v[4:11]))
