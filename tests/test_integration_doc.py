"""INTEGRATION.md shows the bindings a Reef / nova-snark maintainer would add (`extern "C"` blocks, `#[repr(C)]` structs).  They are
the reference-side half of the drop-in boundary (include/reef_msm.h), so they must not drift from the header: every Rust declaration
is parsed and compared with the C prototype of the same name -- number of arguments, pointer / integer / bool kinds and widths, the
return type -- and every `#[repr(C)]` struct with the header's struct, field for field."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_KINDS = {"size_t": "usize", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "bool": "bool",
           "reef_status": "i32", "void": "void"}


def _strip_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def _split_top(args):
    parts, depth, cur = [], 0, ""
    for ch in args:
        depth += ch in "[(<"
        depth -= ch in "])>"
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    return [p for p in parts + [cur] if p.strip()]


def _header_prototypes():
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "reef_msm.h")).read())
    protos = {}
    for m in re.finditer(r"\b([\w ]+?[\s\*]+)((?:reef_|mult_pippenger_)\w+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()

        def kind(a):
            a = a.strip()
            if "*" in a or a.endswith("]"):
                return "ptr"
            return C_KINDS[a.rsplit(" ", 1)[0].replace("const", "").strip()]
        protos[name] = ("ptr" if "*" in ret else C_KINDS[ret], [kind(a) for a in args.split(",")] if args not in ("", "void") else [])
    return protos


def _rust_kind(t):
    t = t.strip()
    return "ptr" if t.startswith("*") else t


def test_every_rust_declaration_matches_the_header():
    protos = _header_prototypes()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    seen = 0
    for m in re.finditer(r"\bfn\s+((?:reef_|mult_pippenger_)\w+)\s*\(([^;{}]*?)\)\s*(?:->\s*([^;{]+?))?\s*;", doc, flags=re.S):
        name, args, ret = m.group(1), _strip_comments(m.group(2)), (m.group(3) or "void").strip()
        assert name in protos, f"INTEGRATION.md binds {name}, which include/reef_msm.h does not declare"
        got = (_rust_kind(ret), [_rust_kind(a.split(":", 1)[1]) for a in _split_top(args)])
        assert got == protos[name], f"{name}: INTEGRATION.md declares {got}, the header {protos[name]}"
        seen += 1
    assert seen >= 20            # the drop-in pair, the handle API, device groups, sum-check, document polynomial, Merkle, key derivation


def _c_struct_fields(name):
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "reef_msm.h")).read())
    body = re.search(r"typedef struct\s*\{([^{}]*)\}\s*" + name + r"\s*;", hdr).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype = decl.rsplit(" ", 1)[0] if "," not in decl else decl.split(",")[0].rsplit(" ", 1)[0]
        for part in decl[len(ctype):].split(","):
            fields.append(re.sub(r"[\*\s]|\[.*?\]", "", part))
    return fields


def test_repr_c_structs_list_the_header_fields_in_order():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    pairs = {"ReefMsmOpts": "reef_msm_opts", "ReefGroupOpts": "reef_msm_group_opts", "ReefPoseidonParams": "reef_poseidon_params",
             "ReefKeygenParams": "reef_keygen_params"}
    for rust, c in pairs.items():
        body = re.search(r"#\[repr\(C\)\]\s*(?:pub\s+)?struct\s+" + rust + r"\s*\{(.*?)\}", doc, flags=re.S).group(1)
        names = [f.split(":")[0].strip() for f in _split_top(_strip_comments(body))]
        assert names == _c_struct_fields(c), f"{rust}: {names} against {c}: {_c_struct_fields(c)}"


def test_the_ctypes_binding_matches_the_header():
    """reef_amd/_ffi.py is how every GPU test, the soak and bench.py reach the library: an argument of the wrong width there would be a bug
    in the checker, not in the product.  Every function the binding declares is compared with the header's prototype."""
    import ctypes

    from reef_amd import _ffi
    lib = _ffi.load()                                    # loads without a GPU (tests/test_abi.py)
    protos = _header_prototypes()

    def kind(t):
        if t is None:
            return "void"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents"):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_uint32: "u32", ctypes.c_uint64: "u64", ctypes.c_bool: "bool"}[t]   # c_size_t IS c_uint64 here
    bound = 0
    for name, (ret_, args_) in protos.items():
        want = (ret_.replace("usize", "u64"), [a.replace("usize", "u64") for a in args_])
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue                                     # declared in the header, not used from Python
        got = (kind(fn.restype), [kind(t) for t in fn.argtypes])
        assert got == want, f"{name}: _ffi.py declares {got}, the header {want}"
        bound += 1
    assert bound >= 60
