#!/usr/bin/env python3
"""Developer aid: builds a copy of the HIP library whose growth kernel carries
cycle probes (s_memtime) between the phases of one Broad() call, and prints
their per-call averages with BS_DEBUG=1.  The product source is not touched:
the probes are patched into a temporary copy of bs_grow_spec.hip.

  python tools/probe_grow_kernel.py buildingsegment_amd/ab/prof.so
  BS_DEBUG=1 BS_LIB_PATH=$PWD/buildingsegment_amd/ab/prof.so python bench.py --steps 1 --warmup 0 ...

Every probe costs ~220 counter units itself (scalar memory round trip); the
"wait" phase (an explicit s_waitcnt right after the plane-state arithmetic)
shows that floor when the gather's latency is already hidden.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "buildingsegment_amd", "csrc", "bs_grow_spec.hip")

PATCHES = [
    ('// wave-cooperative "realloc"',
     '__device__ unsigned long long g_prof[32];\n#define PROBE(i) do { const long long _t = clock64(); pacc[i] += _t - tlast; '
     'tlast = _t; } while (0)\n\n// wave-cooperative "realloc"'),
    ('    auto step = [&](auto mode) -> int {\n      if (__builtin_expect(sp == 0, 0))\n        return 2;\n'
     '      if (__builtin_expect(++iters > iter_cap, 0)) {\n        status = ST_WATCHDOG;\n        return 2;\n      }\n',
     '    long long pacc[16] = {0};\n    long long ncalls = 0, nexp = 0;\n    long long tlast = clock64();\n'
     '    auto step = [&](auto mode) -> int {\n      if (__builtin_expect(sp == 0, 0))\n        return 2;\n'
     '      if (__builtin_expect(++iters > iter_cap, 0)) {\n        status = ST_WATCHDOG;\n        return 2;\n      }\n'
     '      ncalls++;\n      PROBE(7);\n'),
    ('      if constexpr (decltype(mode)::value == 1)\n', '      PROBE(0);\n      if constexpr (decltype(mode)::value == 1)\n'),
    ('          update_state();\n      }\n      bool geo = false;',
     '          update_state();\n      }\n      PROBE(1);\n      __builtin_amdgcn_s_waitcnt(0x0F70);\n      PROBE(2);\n      bool geo = false;'),
    ('      const int last = gstar >= 0 ? gstar : ngv - 1;  // calls 0..last are consumed\n',
     '      PROBE(3);\n      const int last = gstar >= 0 ? gstar : ngv - 1;  // calls 0..last are consumed\n'),
    ('      if (__builtin_expect(gstar < 0, 0))\n        return 0;\n',
     '      PROBE(4);\n      if (__builtin_expect(gstar < 0, 0))\n        return 0;\n      nexp++;\n'),
    ('      ln += cnt;\n      need_state = true;', '      ln += cnt;\n      PROBE(5);\n      need_state = true;'),
    ('        lds_lo = sp - LDS_STACK;  // older entries were overwritten in LDS (still in HBM)\n      return 1;\n    };\n',
     '        lds_lo = sp - LDS_STACK;  // older entries were overwritten in LDS (still in HBM)\n      PROBE(6);\n      return 1;\n    };\n'),
    ('      while (step(std::integral_constant<int, 2>{}) != 2) {\n      }\n    }\n',
     '      while (step(std::integral_constant<int, 2>{}) != 2) {\n      }\n    }\n'
     '    if (lane == 0 && ln > 20000) {\n      for (int i = 0; i < 8; i++)\n        atomicAdd(&g_prof[i], (unsigned long long)pacc[i]);\n      for (int i = 8; i < 12; i++)\n        atomicAdd(&g_prof[12 + i], (unsigned long long)pacc[i]);\n'
     '      atomicAdd(&g_prof[8], (unsigned long long)ncalls);\n      atomicAdd(&g_prof[9], (unsigned long long)nexp);\n    }\n'),
    ("      // settle last call's optimistic claims: the tag must carry my seed\n",
     "      PROBE(8);\n      // settle last call's optimistic claims: the tag must carry my seed\n"),
    ("      // side-effect free classification\n", "      PROBE(9);\n      // side-effect free classification\n"),
    ("      // ---- walk the pending calls in order: consume empty ones, stop at the first that accepts ----\n",
     "      PROBE(10);\n      // ---- walk the pending calls in order: consume empty ones, stop at the first that accepts ----\n"),
    ("      const int rank = __popcll(am & ((1ull << lane) - 1ull));\n",
     "      PROBE(11);\n      const int rank = __popcll(am & ((1ull << lane) - 1ull));\n"),
    ('    if (dbg) {\n      int cnt[6]',
     '    if (dbg) {\n      unsigned long long hp[32];\n      hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof), sizeof(hp));\n'
     '      fprintf(stderr, "[prof] steps=%llu expansions=%llu | src+issue=%llu state=%llu wait=%llu geo+walk=%llu bookA=%llu listS=%llu '
     'push=%llu top=%llu (counter units per step, cumulative over rounds)\\n", hp[8], hp[9], hp[0] / (hp[8] + 1), hp[1] / (hp[8] + 1), '
     'hp[2] / (hp[8] + 1), hp[3] / (hp[8] + 1), hp[4] / (hp[8] + 1), hp[5] / (hp[8] + 1), hp[6] / (hp[8] + 1), hp[7] / (hp[8] + 1));\n'
     '      fprintf(stderr, "[prof2] geo=%llu settle=%llu classify=%llu claims+walk=see geo+walk slabcheck=%llu (geo+walk above = claims and walk only; listS = after the slab check)\\n", hp[20] / (hp[8] + 1), hp[21] / (hp[8] + 1), hp[22] / (hp[8] + 1), hp[23] / (hp[8] + 1));\n'
     '      int cnt[6]'),
]


def main() -> int:
    out = os.path.abspath(sys.argv[1])
    s = open(SRC).read()
    for old, new in PATCHES:
        if old not in s:
            print("anchor not found (kernel changed? update tools/probe_grow_kernel.py):\n" + old, file=sys.stderr)
            return 1
        s = s.replace(old, new, 1)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        v = os.path.join(td, "bs_grow_spec_probed.hip")
        open(v, "w").write(s)
        subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), v, out])
    print(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
