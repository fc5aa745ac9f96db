"""Parses the C ABI out of include/bevyray_amd.h and the Rust FFI declarations out of integration/bevyray_amd_sys/src/lib.rs
into one comparable form (tests/test_abi_binding.py: no Rust toolchain exists here, so nothing else checks that crate)."""
import re

C_TO_RUST = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64", "void": "c_void",
             "char": "c_char", "brt_ctx": "brt_ctx", "brt_stats": "brt_stats"}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//.*", "", text)


def _rust_type(c_type):
    t = " ".join(c_type.replace("*", " * ").split())
    const = t.startswith("const ")
    if const:
        t = t[6:]
    stars, base = t.count("*"), t.replace("*", "").strip()
    r = C_TO_RUST[base]
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def parse_header(path):
    """-> {"functions": {name: (ret, [(arg, type)])}, "constants": {name: int}, "stats": [(field, type)], "abi": int}"""
    h = _strip_c(open(path).read())
    body = h[h.index('extern "C" {'):]
    fns = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?)\s*(brt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", body, flags=re.M | re.S):
        al = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a.strip())
                al.append((m.group(2), _rust_type(m.group(1).strip())))
        fns[name] = (_rust_type(ret.strip()), al)
    consts = {}
    for name, val in re.findall(r"#define\s+(BRT_[A-Z0-9_]+)\s+(\d+)u?", h):
        consts[name] = int(val)
    for block in re.findall(r"enum\s*\{(.*?)\}", h, flags=re.S):
        nxt = 0
        for item in block.split(","):
            item = item.strip()
            if not item:
                continue
            m = re.match(r"(BRT_[A-Z0-9_]+)\s*(?:=\s*(-?\d+)u?)?$", item)
            if not m:
                continue
            nxt = int(m.group(2)) if m.group(2) is not None else nxt
            consts[m.group(1)] = nxt
            nxt += 1
    st = re.search(r"typedef struct brt_stats \{(.*?)\} brt_stats;", h, flags=re.S).group(1)
    stats = [(n, C_TO_RUST[t]) for t, n in re.findall(r"(uint64_t|uint32_t|double|float)\s+([a-z_]+)\s*;", st)]
    return {"functions": fns, "constants": consts, "stats": stats, "abi": consts["BRT_ABI_VERSION"]}


def parse_rust(path):
    src = re.sub(r"//.*", "", open(path).read())
    fns = {}
    ext = src[src.index('extern "C" {'):]
    ext = ext[:ext.index("\n}")]
    for name, args, ret in re.findall(r"pub fn (brt_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        al = []
        for a in [x.strip() for x in " ".join(args.split()).split(",") if x.strip()]:
            n, t = a.split(":", 1)
            al.append((n.strip(), " ".join(t.split())))
        fns[name] = (" ".join((ret or "()").split()), al)
    consts = {n: int(v) for n, v in re.findall(r"pub const (BRT_[A-Z0-9_]+)\s*:\s*[iu]32\s*=\s*(-?\d+)\s*;", src)}
    st = re.search(r"pub struct brt_stats \{(.*?)\n\}", src, flags=re.S).group(1)
    stats = [(n, t) for n, t in re.findall(r"pub ([a-z_]+)\s*:\s*(u64|u32|f64|f32)", st)]
    return {"functions": fns, "constants": consts, "stats": stats, "abi": consts.get("BRT_ABI_VERSION")}
