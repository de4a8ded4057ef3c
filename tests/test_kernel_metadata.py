"""Code-object metadata of every kernel in the product library (hipcc cross-compiles gfx950 without a GPU): no kernel
may use scratch memory (a spill inside these kernels has always meant a register-budget regression — the scan keeps 64
query-row registers live), and the throughput kernels must fit the 6-waves-per-SIMD budget of 80 VGPRs."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slam-loop-closing_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def kernels_of(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip",
                           src, "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    meta = text[text.index("amdhsa.kernels:"):]
    ks = {}
    for block in re.split(r"\n  - \.a", meta)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        ks[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                    for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size")}
    return ks


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["lcm_kernels.hip", "lcm_mfma.hip"])
def test_no_kernel_uses_scratch_and_budgets_hold(src, tmp_path):
    ks = kernels_of(os.path.join(CSRC, src), tmp_path)
    assert len(ks) >= 5
    for name, m in ks.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0, (name, m)
        assert m["group_segment_fixed_size"] <= 65536, (name, m)
    if src == "lcm_kernels.hip":
        # k_score_rowlane<THREADS, QPT, ARGMIN_MODE, WRITE_KEYS, PACKED>: bulk / online kernels (modes 0, 1) on the 80-VGPR
        # budget, the 6-rows-per-lane A/B on 64, the pair-mode key kernel (mode 2) on 96
        seen = 0
        for name, m in ks.items():
            t = re.search(r"k_score_rowlaneILi(\d+)ELi(\d+)ELi(\d)ELb([01])ELb([01])E", name)
            if not t:
                continue
            seen += 1
            qpt, mode, packed = int(t.group(2)), int(t.group(3)), t.group(5) == "1"
            limit = 96 if mode == 2 else (64 if (packed and qpt == 6) else 80)
            assert m["vgpr_count"] <= limit, (name, m)
            if packed:                       # lane-private LDS words: 6 (8) workgroups of it must fit a CU's 160 KB
                assert m["group_segment_fixed_size"] * (8 if qpt == 6 else 6) <= 160 * 1024, (name, m)
        assert seen >= 20


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_inner_loop_is_priority_steered(tmp_path):
    """The headline kernel's instruction stream as the assembler emits it: every v_bcnt_u32_b32 and v_min3_u32 of the scan
    runs between an `s_setprio 3` and the next `s_setprio 0`, every v_xor_b32 of the scan outside (the other
    waves' half-rate xors issue beside this wave's quarter-rate popcounts — 36 instead of 54 SIMD-cycles per 64
    distances).  A compiler or source change that moves one of them across is a 15-30 % regression no parity test sees."""
    out = tmp_path / "lcm_kernels.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip",
                           os.path.join(CSRC, "lcm_kernels.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    for mode in (0, 1):                                  # distance-only and argmin kernels of the packed route
        sym = f"_ZN3lcm15k_score_rowlaneILi256ELi8ELi{mode}ELb0ELb1EEEvNS_9ScoreArgsE"
        body = text[text.index(sym + ":"):]
        body = body[:body.index("s_endpgm")]
        prio, counts = 0, {"bcnt_hi": 0, "bcnt_lo": 0, "xor_hi": 0, "xor_lo": 0, "min3_hi": 0, "min3_lo": 0}
        for line in body.splitlines():
            ins = line.strip().split(" ")[0] if line.strip() else ""
            if ins == "s_setprio":
                prio = int(line.split()[1])
            elif ins.startswith("v_bcnt_u32_b32"):
                counts["bcnt_hi" if prio else "bcnt_lo"] += 1
            elif ins.startswith("v_xor_b32"):
                counts["xor_hi" if prio else "xor_lo"] += 1
            elif ins.startswith("v_min3_u32"):
                counts["min3_hi" if prio else "min3_lo"] += 1
        # the scan: 2 buffers x 8 query rows x 16 (xor, bcnt) + 1 min3; the argmin re-scan (< 1 % of the work, unrolled over the
        # 8 query rows) adds 8 x 8 xor / bcnt at low priority
        assert counts["bcnt_hi"] == 256 and counts["xor_lo"] >= 256 and counts["min3_hi"] == 16, (mode, counts)
        assert counts["xor_hi"] == 0 and counts["min3_lo"] == 0, (mode, counts)
        assert counts["bcnt_lo"] <= (64 if mode == 1 else 0), (mode, counts)
