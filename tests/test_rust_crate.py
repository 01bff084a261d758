"""rust/headroom-hip is shipped as source only (no Rust toolchain in the image): these checks keep it from rotting.
The extern block must list exactly the symbols include/headroom_hip.h declares (= what the library exports), with the
same number of arguments; every #[repr(C)] struct must have the header's fields in the header's order with a matching
scalar type; the error / format / mode constants must carry the header's values."""
import os
import re

import headroom_amd as hr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "headroom_hip.h")).read()
LIB_RS = open(os.path.join(ROOT, "rust", "headroom-hip", "src", "lib.rs")).read()

C_TO_RUST = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint16_t": "u16", "double": "f64", "float": "f32",
             "int": "c_int", "size_t": "usize"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_functions():
    text = strip_comments(HEADER)
    out = {}
    for m in re.finditer(r"^[\w \*]+?\b(hr_\w+)\(([^;{]*?)\);", text, flags=re.M | re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def rust_functions():
    block = LIB_RS[LIB_RS.index('extern "C" {'):]
    block = block[:block.index("\n    }\n")]
    out = {}
    for m in re.finditer(r"pub fn (hr_\w+)\((.*?)\)(?: -> [^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def c_structs():
    text = strip_comments(HEADER)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"(const )?(\w+) (\*?)(.*)$", decl)
            ctype, ptr, names = mm.group(2), mm.group(3), mm.group(4)
            for name in names.split(","):
                name = name.strip()
                is_ptr = bool(ptr) or name.startswith("*")
                name = name.lstrip("*")
                dims = re.findall(r"\[(\w+)\]", name)
                name = re.sub(r"\[.*", "", name)
                fields.append((name, ctype, is_ptr, dims))
        out[m.group(1)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+) \{(.*?)\n    \}", LIB_RS, flags=re.S):
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^\n]+?),\n", m.group(2) + "\n")]
        out[m.group(1)] = fields
    return out


def test_extern_block_lists_every_export_with_the_headers_arity():
    c, r = c_functions(), rust_functions()
    assert set(c) == set(hr.EXPORTS), set(c) ^ set(hr.EXPORTS)       # the header declares what the library exports
    assert set(r) == set(c), set(r) ^ set(c)
    for name in c:
        assert r[name] == c[name], (name, r[name], c[name])


def test_repr_c_structs_follow_the_header_field_for_field():
    c, r = c_structs(), rust_structs()
    for name in ("hr_config", "hr_track", "hr_result", "hr_shard_edge", "hr_shard_partial", "hr_pcm_info", "hr_tp_target", "hr_analysis"):
        assert name in c and name in r, name
        cf, rf = c[name], r[name]
        assert [f[0] for f in cf] == [f[0] for f in rf], (name, cf, rf)
        for (fname, ctype, is_ptr, dims), (_, rtype) in zip(cf, rf):
            if is_ptr:
                assert rtype.startswith("*const") or rtype.startswith("*mut"), (name, fname, rtype)
                continue
            base = C_TO_RUST[ctype]
            want = base
            for d in reversed(dims):
                want = f"[{want}; {d}]"
            assert rtype == want, (name, fname, rtype, want)


def test_constants_carry_the_headers_values():
    text = strip_comments(HEADER)
    values = {}
    for m in re.finditer(r"\b(HR_\w+) = (-?\d+)u?", text):
        values[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum \{([^}]*)\}", text):  # implicit enumerators: previous + 1
        prev = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                prev = int(v.rstrip("u"))
            else:
                name, prev = item, prev + 1
            values[name] = prev
    for m in re.finditer(r"#define (HR_\w+) (\d+)u?", text):
        values[m.group(1)] = int(m.group(2))
    rust = {m.group(1): int(m.group(2).replace("_", "")) for m in re.finditer(r"pub const (HR_\w+): \w+ = (-?[\d_]+);", LIB_RS)}
    assert len(rust) >= 30
    for name, v in rust.items():
        assert name in values and values[name] == v, (name, v, values.get(name))
    for must in ("HR_ERR_NOT_READY", "HR_FMT_S24", "HR_GATE_LOUDNORM", "HR_TP_SAMPLE", "HR_GAIN_NONE", "HR_KIND_AAC", "HR_MAX_SAMPLE_RATE"):
        assert must in rust, must


def test_config_default_is_written_by_hand():
    # a raw pointer field has no Default: the sys struct must not derive it, the safe HrConfig implements it
    m = re.search(r"#\[derive\(([^)]*)\)\]\s*pub struct hr_config", LIB_RS)
    assert m and "Default" not in m.group(1)
    assert "impl Default for HrConfig" in LIB_RS
