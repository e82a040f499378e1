"""Mechanical check of the Rust drop-in crate source (bindings/rust/) -- there is no rustc in the image,
so the `extern "C"` block is parsed and compared with include/qoaudio_amd.h declaration by declaration: symbol exists,
same number of arguments, same pointer depth / constness, same integer width, same return type; the #[repr(C)] structs
against the header's struct fields; and the public API surface against the reference's signatures that matter to
callers (src/lib.rs:180-188 return references)."""
import os
import re

from conftest import ROOT

RS = os.path.join(ROOT, "bindings", "rust", "src", "qoaudio_mi355x.rs")
HDR = os.path.join(ROOT, "include", "qoaudio_amd.h")

RUST_SCALARS = {"f64": "double", "c_void": "void", "i64": "int64_t", "u8": "uint8_t", "u16": "uint16_t", "u32": "uint32_t", "u64": "uint64_t", "i16": "int16_t", "i32": "int32_t",
                "usize": "size_t", "c_int": "int", "c_double": "double", "c_char": "char"}


def rust_type_to_c(t):
    """'*mut *mut qoa_encoder' -> ('qoa_encoder', ['mut', 'mut'])  (pointer levels outermost first)"""
    t = t.strip()
    levels = []
    while t.startswith("*"):
        m = re.match(r"\*(mut|const)\s+(.*)", t)
        levels.append(m.group(1))
        t = m.group(2).strip()
    return RUST_SCALARS.get(t, t), levels


def c_type(t):
    """'const qoa_desc *desc' / 'uint8_t out[8]' -> (base, levels outermost first), name dropped"""
    t = re.sub(r"/\*.*?\*/", "", t).strip()
    arr = re.search(r"\[\d*\]\s*$", t)
    if arr:
        t = t[:arr.start()].strip()
    m = re.match(r"(.*?)([A-Za-z_]\w*)?$", t)
    body = m.group(1).strip() if m.group(2) and (m.group(1).strip().endswith("*") or " " in m.group(1).strip()) else t
    # body now has no parameter name unless the type was a single word
    toks = body.replace("*", " * ").split()
    base_const = False
    base = None
    levels = []
    for tok in toks:
        if tok == "const":
            if base is None:
                base_const = True
            else:
                if levels:
                    levels[-1] = ("ptr", "constptr")
        elif tok == "*":
            levels.append(("ptr", None))
        else:
            if base is None:
                base = tok
            elif tok in ("int", "char", "long", "unsigned"):
                base += " " + tok
    if arr:
        levels.append(("ptr", None))
    # level i (innermost = 0) is `*const` in Rust terms when what it points at is const: the base type for the innermost
    # pointer, a `*const` pointer (C: `* const`) for the levels outside it
    out = []
    for i in range(len(levels)):
        inner_const = base_const if i == 0 else levels[i - 1][1] == "constptr"
        out.append("const" if inner_const else "mut")
    return base, list(reversed(out))  # outermost first


def parse_header():
    text = open(HDR).read()
    text = re.sub(r"//.*", "", text)
    decls = {}
    for m in re.finditer(r"^([A-Za-z_][\w \*]*?)\b(qoa_\w+)\s*\(([^;{]*?)\)\s*;", text, re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        decls[name] = (ret, arglist)
    return decls


def parse_extern_block():
    text = open(RS).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n    }\n")]
    fns = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, re.S):
        args = " ".join(m.group(2).split())
        arglist = [] if not args else [a.split(":", 1)[1].strip() for a in args.split(", ")]
        fns[m.group(1)] = (m.group(3).strip() if m.group(3) else None, arglist)
    return fns


def test_every_extern_symbol_matches_the_c_header():
    hdr, rs = parse_header(), parse_extern_block()
    assert len(rs) >= 24
    for name in ("qoa_decoder_new_reader", "qoa_multi_create", "qoa_multi_free", "qoa_multi_shard", "qoa_multi_encode",
                 "qoa_multi_decode", "qoa_encode_batch_multi", "qoa_decode_batch_multi"):
        assert name in rs, name
    for name, (ret, args) in rs.items():
        assert name in hdr, f"{name} is not declared in include/qoaudio_amd.h"
        c_ret, c_args = hdr[name]
        assert len(args) == len(c_args), (name, args, c_args)
        if ret is None:
            assert c_ret == "void", (name, c_ret)
        else:
            assert rust_type_to_c(ret) == c_type(c_ret + " x"), (name, ret, c_ret)
        for a, c in zip(args, c_args):
            assert rust_type_to_c(a) == c_type(c), (name, a, c)


def test_every_header_symbol_is_bound():
    """Completeness (round 5): every function include/qoaudio_amd.h declares -- the `qoa_debug_*` test hooks excepted -- must be
    in the shim's `extern "C"` block; round 4's shim bound the multi-GPU entry points but not the single-GPU plans, the host
    pipeline or qoa_stream_scan, i.e. not the entry points through which a Rust host reaches the kernels' rate."""
    hdr, rs = parse_header(), parse_extern_block()
    assert len(hdr) >= 55, sorted(hdr)
    missing = sorted(n for n in hdr if not n.startswith("qoa_debug_") and n not in rs)
    assert not missing, missing
    unknown = sorted(n for n in rs if n not in hdr)
    assert not unknown, unknown
    text = open(RS).read()
    for wrapper in ("pub struct EncodePlan", "pub struct DecodePlan", "pub struct HostPipeline", "pub fn stream_scan(",
                    "pub fn wave_stats(", "pub fn launch_shape(", "pub fn read_samples(", "pub fn lms_state("):
        assert wrapper in text, wrapper


def test_every_symbol_the_shim_binds_is_exported(tmp_path):
    import ctypes
    so = os.path.join(ROOT, "qoaudio_amd", "libqoaudio_amd.so")
    lib = ctypes.CDLL(so)
    for name in parse_extern_block():
        getattr(lib, name)


def _struct_fields_c(name):
    text = open(HDR).read()
    m = re.search(r"typedef struct (?:\w+\s*)?\{([^}]*)\}\s*" + name + r"\s*;", text, re.S)
    assert m, name
    body = re.sub(r"//.*|/\*.*?\*/", "", m.group(1), flags=re.S)
    consts = {m.group(1): m.group(2) for m in re.finditer(r"#define (QOA_\w+) (\d+)\b", text)}
    fields = []
    for f in body.split(";"):
        if not f.strip():
            continue
        ctype, names = " ".join(f.split()[:-1]), f.split()[-1]
        if "," in f:  # `double a, b, c`
            ctype = f.split()[0]
            names = f.strip()[len(ctype):]
        for name in names.split(","):
            name = name.strip()
            arr = re.match(r"(\w+)\[(\w+)\]$", name)
            if arr:
                fields.append((arr.group(1), f"{ctype}[{consts.get(arr.group(2), arr.group(2))}]"))
            else:
                fields.append((name, ctype))
    return fields


def _struct_fields_rs(name):
    text = open(RS).read()
    m = re.search(r"pub struct " + name + r" \{(.*?)\}", text, re.S)
    body = m.group(1)
    out = []
    for n, t in re.findall(r"pub (\w+): (\[[^\]]+\]|[\w:]+)", body):
        arr = re.match(r"\[(\w+); (\d+)\]", t)
        if arr:  # [u64; 8] -> ("uint64_t", 8) as the C side's `uint64_t name[8]`
            out.append((n, RUST_SCALARS.get(arr.group(1), arr.group(1)) + f"[{arr.group(2)}]"))
        else:
            out.append((n, RUST_SCALARS.get(t.strip(), t.strip())))
    return out


def test_repr_c_structs_match_the_header():
    for name in ("qoa_desc", "qoa_frame_header", "qoa_mode", "qoa_item", "qoa_clip", "qoa_launch_shape", "qoa_wave_stats",
                 "qoa_stream_info"):
        c = [(n, t) for n, t in _struct_fields_c(name)]
        rs = _struct_fields_rs(name)
        assert [n for n, _ in c] == [n for n, _ in rs], (name, c, rs)
        for (_, ct), (_, rt) in zip(c, rs):
            assert ct == rt, (name, ct, rt)
    text = open(RS).read()
    for name in ("qoa_desc", "qoa_frame_header", "qoa_mode", "qoa_item", "qoa_clip", "qoa_launch_shape", "qoa_wave_stats",
                 "qoa_stream_info"):
        assert re.search(r"#\[repr\(C\)\]\s*(#\[derive\([^)]*\)\]\s*)?pub struct " + name, text), name


def test_public_signatures_equal_the_reference():
    """The reference's signatures, restated (src/lib.rs:127, :172, :180, :186, :190, :335, :367, :404, :418, :619, :632, :641,
    :713, :749, :760): what a caller's code compiles against."""
    text = " ".join(open(RS).read().split())
    for sig in (
        "pub fn new(mut reader: R) -> Result<Self, DecodeError>",
        "pub fn into_inner(mut self) -> R",
        "pub fn mode(&self) -> &ProcessingMode",
        "pub fn current_frame_header(&self) -> &FrameHeader",
        "pub fn total_duration(&self) -> Option<Duration>",
        "pub fn open<P: AsRef<Path>>(path: P) -> Result<QoaDecoder<io::BufReader<File>>, DecodeError>",
        "pub fn new_streaming() -> Result<Self, DecodeError>",
        "pub fn decode_frame(&mut self, frame_data: &[u8]) -> Result<Vec<i16>, DecodeError>",
        "impl<R: io::Read> Iterator for QoaDecoder<R>",
        "type Item = Result<QoaItem, DecodeError>;",
        "pub fn new(desc: &QoaDesc) -> Result<Self, EncodeError>",
        "pub fn encode(&mut self, sample_data: &[i16]) -> Result<Vec<u8>, EncodeError>",
        "pub fn write_header<W: io::Write>(&self, writer: &mut W) -> Result<(), EncodeError>",
        "pub fn encode_frame<W: io::Write>(&mut self, sample_data: &[i16], writer: &mut W) -> Result<usize, EncodeError>",
        "pub fn encode_all(sample_data: &[i16], desc: &QoaDesc) -> Result<Vec<u8>, EncodeError>",
        "pub fn decode_all<R: io::Read>(mut reader: R) -> Result<DecodedQoa, DecodeError>",
        "pub fn open_and_decode_all<P: AsRef<Path>>(path: P) -> Result<DecodedQoa, DecodeError>",
    ):
        assert sig in text, sig
    # (`into_inner(mut self)` / `decode_all(mut reader)`: `mut` on a by-value binding is not part of a signature)


def test_integration_md_excerpts_are_verbatim():
    """INTEGRATION.md section 2 quotes the shim instead of restating it: every ```rust block there must be a verbatim run of
    lines of bindings/rust/src/qoaudio_mi355x.rs (the document went stale once when it carried its own copy)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rs = open(RS).read()
    blocks = re.findall(r"```rust\n(.*?)\n```", doc, re.S)
    assert len(blocks) >= 5
    for b in blocks:
        assert b in rs, b[:200]
