#!/usr/bin/env python3
"""Build variants of the library with single LDS access sites of the attention backward removed
(wrong results on purpose) to attribute SQ_LDS_BANK_CONFLICT; experiments only."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visiontransformer_amd", "csrc")
def build(name, edit):
    d = "/tmp/var_%s/a/b" % name
    shutil.rmtree("/tmp/var_%s" % name, ignore_errors=True); shutil.copytree(SRC, d); shutil.copytree(os.path.join(ROOT, "include"), "/tmp/var_%s/include" % name)
    p = os.path.join(d, "attn_bwd.hip"); s = open(p).read(); s2 = edit(s); assert s2 != s or name == "base", name; open(p, "w").write(s2)
    os.makedirs(os.path.join(ROOT, "visiontransformer_amd", "variants"), exist_ok=True)
    objs = []
    procs = []
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip"):
            o = os.path.join(d, f + ".o"); objs.append(o)
            procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-w", "-c", os.path.join(d, f), "-o", o]))
    for pr in procs: assert pr.wait() == 0
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "visiontransformer_amd", "variants", "libdswin_%s.so" % name)] + objs)
    print("built", name)
V = {
 "base": lambda s: s,
 "noDQst": lambda s: s.replace("            if (ee < d) {\n#pragma unroll\n                for (int r = 0; r < 4; ++r) io<T>::st(dqkv + trow(qb * 32 + 16 * mq + 4 * l4 + r) * C3 + qoff + ee, v[r] * scale);", "            if (ee < d && scale == 12345.f) {\n#pragma unroll\n                for (int r = 0; r < 4; ++r) io<T>::st(dqkv + trow(qb * 32 + 16 * mq + 4 * l4 + r) * C3 + qoff + ee, v[r] * scale);"),
 "noDKVst": lambda s: s.replace("                if (2 * dw < d) {\n                    T *dst = dqkv + trow(32 * tile + row) * C3 + (which == 0 ? koff : voff) + 2 * dw;", "                if (2 * dw < d && scale == 12345.f) {\n                    T *dst = dqkv + trow(32 * tile + row) * C3 + (which == 0 ? koff : voff) + 2 * dw;"),
 "noBGws": lambda s: s.replace("                dst[it] = acc;\n            }\n        } else {\n            const int dj = (tid % (2 * WS)) - (WS - 1);", "                if (scale == 12345.f) dst[it] = acc;\n            }\n        } else {\n            const int dj = (tid % (2 * WS)) - (WS - 1);"),
 "noKt": lambda s: s.replace("Kl[e * KTS + row] = (T)(k2 & 0xffffu);", "if (row == 99999) Kl[e * KTS + row] = (T)(k2 & 0xffffu);").replace("Kl[(e + 1) * KTS + row] = (T)(k2 >> 16);", "if (row == 99999) Kl[(e + 1) * KTS + row] = (T)(k2 >> 16);"),
 "noDsW": lambda s: s.replace("*reinterpret_cast<uint2 *>(dsrow + 4 * ((c0 & 4) | ((c0 & 3) ^ fk))) = make_uint2(db.x, db.y);", "if (fk == 99) *reinterpret_cast<uint2 *>(dsrow) = make_uint2(db.x, db.y);").replace("*reinterpret_cast<uint2 *>(dsrow + 4 * ((c1 & 4) | ((c1 & 3) ^ fk))) = make_uint2(db.z, db.w);", "if (fk == 98) *reinterpret_cast<uint2 *>(dsrow) = make_uint2(db.z, db.w);"),
 "noTrQG": lambda s: s.replace("const uint2 g0 = lds_read_tr16(gtr + (16 * s) * QS + 32 * u), g1 = lds_read_tr16(gtr + (16 * s + 8) * QS + 32 * u);", "const uint2 g0 = make_uint2(pb.x, pb.y), g1 = make_uint2(pb.z, pb.w);").replace("const uint2 q0 = lds_read_tr16(qtr + (16 * s) * QS + 32 * u), q1 = lds_read_tr16(qtr + (16 * s + 8) * QS + 32 * u);", "const uint2 q0 = make_uint2(db.x, db.y), q1 = make_uint2(db.z, db.w);"),
 "noTrDs": lambda s: s.replace("const uint2 a0 = lds_read_tr16(abase + (32 * s) * DSS), a1 = lds_read_tr16(abase + (32 * s + 16) * DSS);", "const uint2 a0 = make_uint2(s, lane), a1 = make_uint2(lane, s);"),
 "noEpi": lambda s: s.replace("        T *slice = dSs + wave * (32 * QS);", "        if (scale != 12345.f) return;\n        T *slice = dSs + wave * (32 * QS);"),
 "noInit": lambda s: s.replace("float v0 = nlse_s[qb * 32 + q_c + 4 * hh];", "float v0 = 0.001f * qb;").replace("if constexpr (BIAS) v0 += tb[lane_base + (q_c / WS) * TS + (q_c % WS)];", "").replace("pacc[r] = ndelta_s[qb * 32 + q_c + 4 * hh];", "pacc[r] = 0.002f * qb;"),
 "noStore": lambda s: s.replace("*reinterpret_cast<uint32_t *>(Qb + row * QS + e) = q2;", "if (row == 9999) *reinterpret_cast<uint32_t *>(Qb + row * QS + e) = q2;").replace("*reinterpret_cast<uint32_t *>(dOb + row * QS + e) = eok ? ng[i] : 0u;", "if (row == 9999) *reinterpret_cast<uint32_t *>(dOb + row * QS + e) = eok ? ng[i] : 0u;"),
 "noShfl": lambda s: s.replace("for (int o = 1; o < DW; o <<= 1) part += __shfl_xor(part, o, 64);", "part *= 1.5f;"),
 "noKrd": lambda s: s.replace("const uint2 b0 = *reinterpret_cast<const uint2 *>(bbase + 32 * s), b1 = *reinterpret_cast<const uint2 *>(bbase + 32 * s + 16);", "const uint2 b0 = make_uint2(s + 1, lane), b1 = make_uint2(lane, s + 2);"),
 "noBgw": lambda s: s.replace("                if (lane < 16) {\n                    float *ent = bgw + wave * BGW_FLOATS + (2 * qb) * BGW_STRIDE + 15 - lane;", "                if (lane < 0) {\n                    float *ent = bgw + wave * BGW_FLOATS + (2 * qb) * BGW_STRIDE + 15 - lane;"),
 "noAop": lambda s: s.replace("const uint4 qa = *reinterpret_cast<const uint4 *>(Qb + l31 * QS + 16 * s + 8 * hh);", "const uint4 qa = kfb[s];").replace("const uint4 ga = *reinterpret_cast<const uint4 *>(dOb + l31 * QS + 16 * s + 8 * hh);", "const uint4 ga = vfb[s];"),
}
for n in (sys.argv[1:] or list(V)):
    build(n, V[n])
