"""Timing variants of the plane-attention FORWARD kernel (results are wrong by construction): each removes one kind of work
so that its cost inside the pipelined kernel can be read off.  Writes scratch/variants/ap_<name>.hip and builds lib files."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "lvt_amd/csrc/attention_pipe.hip")).read()
a = src.index("template <int BT, int BH, int BW, int MASKED, int NCH>\n__device__ __forceinline__ void attn_fwd_body")
b = src.index("template <int BT, int BH, int BW, int MASKED>\n__global__ __launch_bounds__(256, 1) void lvt_attn_fwd_planes_kernel")
fwd = src[a:b]


def variant(name, f):
    out = src[:a] + f(fwd) + src[b:]
    path = os.path.join(root, "scratch/variants/ap_%s.hip" % name)
    open(path, "w").write(out)
    csrc = os.path.join(root, "lvt_amd/csrc")
    obj = path.replace(".hip", ".o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                           "-I" + csrc, "-c", path, "-o", obj])
    others = [os.path.join(csrc, o) for o in os.listdir(csrc) if o.endswith(".o") and o != "attention_pipe.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + others + [obj, "-o",
                           os.path.join(root, "scratch/variants/libattn_%s.so" % name)])
    os.remove(obj)
    print("built", name)


os.makedirs(os.path.join(root, "scratch/variants"), exist_ok=True)
V = {
    "nosoft": lambda s: s.replace("if constexpr (t > 0) softmax_piece(t - 1, s);", "").replace(
        "static_for<8>([&](auto pc) { softmax_piece(NCH - 1, decltype(pc)::value); });", ""),
    "nopstore": lambda s: re.sub(r"#pragma unroll\n            for \(int gq = 2 \* s2; gq < 2 \* s2 \+ 2; \+\+gq\)\n.*?;\n", "", s, flags=re.S),
    "nopark": lambda s: re.sub(r"ap_park<AP_LD[RT], [^;]*?>\(AP_G\(t \+ 1\), nxt, tid\);", ";", s),
    "nopsplit": lambda s: s.replace("""            at_split8(make_float4(st[T][8 * s2 + 0], st[T][8 * s2 + 1], st[T][8 * s2 + 2], st[T][8 * s2 + 3]),
                      make_float4(st[T][8 * s2 + 4], st[T][8 * s2 + 5], st[T][8 * s2 + 6], st[T][8 * s2 + 7]),
                      pb[0], pb[1], pb[2]);""", "            pb[0] = qb[T][0]; pb[1] = qb[T][1]; pb[2] = qb[T][2];"),
    "noph2": lambda s: s.replace("static_for<NCH>([&](auto cc) {\n        constexpr int c = decltype(cc)::value, t = NCH + c;",
                                 "static_for<0>([&](auto cc) {\n        constexpr int c = decltype(cc)::value, t = NCH + c;"),
    "noload": lambda s: s.replace("if constexpr (t + 2 < NIT) load_item(t + 2, AP_G(t));", ""),
}
for n in (sys.argv[1:] or V):
    variant(n, V[n])
