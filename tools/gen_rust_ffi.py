#!/usr/bin/env python3
"""include/spf_hip.h -> the Rust `extern "C"` block a `parasol_runtime` shim binds, so that the binding
cannot drift from the header: every struct, enum constant and function of the header, nothing else.

    python3 tools/gen_rust_ffi.py include/spf_hip.h > include/spf_hip.rs
    python3 tools/gen_rust_ffi.py --check-library spf_amd/lib/libspf_hip.so include/spf_hip.h     (nm -D == the header's functions)

The parser handles exactly the C subset the header uses (opaque structs, two plain structs, enums, function prototypes with
pointer / integer / double arguments); anything else is an error, not a guess."""
import re
import subprocess
import sys

INT_TYPES = {"int": "c_int", "size_t": "usize", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32",
             "double": "f64", "char": "c_char", "void": "c_void", "spf_status": "spf_status"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)          # preprocessor lines
    text = text.replace('extern "C" {', " ")
    return text


def top_level_statements(text: str):
    depth, cur = 0, []
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth < 0:      # the closing brace of extern "C"
                depth = 0
                continue
        if ch == ";" and depth == 0:
            stmt = " ".join("".join(cur).split())
            if stmt:
                yield stmt
            cur = []
        else:
            cur.append(ch)


class Header:
    def __init__(self, path: str):
        self.opaque, self.structs, self.enums, self.functions = [], [], [], []
        self.known = dict(INT_TYPES)
        for stmt in top_level_statements(strip_comments(open(path).read())):
            self._statement(stmt)

    def _statement(self, s: str):
        m = re.fullmatch(r"typedef struct (\w+) (\w+)", s)
        if m:
            self.opaque.append(m.group(2))
            self.known[m.group(2)] = m.group(2)
            return
        m = re.fullmatch(r"typedef int (\w+)", s)
        if m:
            self.known[m.group(1)] = m.group(1)
            return
        m = re.fullmatch(r"typedef struct (\w+) \{(.*)\} (\w+)", s)
        if m:
            fields = []
            for f in m.group(2).split(";"):
                f = f.strip()
                if not f:
                    continue
                fm = re.fullmatch(r"(\w+) ([\w, \[\]]+)", f)
                if not fm or fm.group(1) not in INT_TYPES:
                    raise SystemExit(f"gen_rust_ffi: cannot parse struct field {f!r}")
                for name in fm.group(2).split(","):
                    name = name.strip()
                    am = re.fullmatch(r"(\w+)\[(\d+)\]", name)
                    fields.append((am.group(1), f"[{INT_TYPES[fm.group(1)]}; {am.group(2)}]") if am else (name, INT_TYPES[fm.group(1)]))
            self.structs.append((m.group(3), fields))
            self.known[m.group(3)] = m.group(3)
            return
        m = re.fullmatch(r"(?:typedef )?enum (\w+ )?\{(.*)\}( \w+)?", s)
        if m:
            name = (m.group(3) or "").strip() or None
            consts, nxt = [], 0
            for c in m.group(2).split(","):
                c = c.strip()
                if not c:
                    continue
                cm = re.fullmatch(r"(\w+)(?: = (-?\d+))?", c)
                if not cm:
                    raise SystemExit(f"gen_rust_ffi: cannot parse enum constant {c!r}")
                val = int(cm.group(2)) if cm.group(2) is not None else nxt
                consts.append((cm.group(1), val))
                nxt = val + 1
            self.enums.append((name, consts))
            if name:
                self.known[name] = name
            return
        m = re.fullmatch(r"(.+?)\b(\w+)\((.*)\)", s)
        if m:
            ret = self._type(m.group(1).strip(), returning=True)
            args = []
            body = m.group(3).strip()
            if body and body != "void":
                for i, a in enumerate(body.split(",")):
                    args.append(self._param(a.strip(), i))
            self.functions.append((m.group(2), args, ret))
            return
        raise SystemExit(f"gen_rust_ffi: cannot parse statement {s!r}")

    def _type(self, c: str, returning=False) -> str:
        """a C type without a declarator name"""
        toks = re.findall(r"\w+|\*", c)
        const_base = False
        i = 0
        while toks[i] in ("const", "struct", "enum"):
            const_base |= toks[i] == "const"
            i += 1
        base = toks[i]
        i += 1
        if i < len(toks) and toks[i] == "const":
            const_base = True
            i += 1
        if base not in self.known:
            raise SystemExit(f"gen_rust_ffi: unknown type {base!r} in {c!r}")
        rust = self.known[base]
        prev_const = const_base
        n_ptr = 0
        while i < len(toks):
            if toks[i] != "*":
                raise SystemExit(f"gen_rust_ffi: cannot parse type {c!r}")
            rust = ("*const " if prev_const else "*mut ") + rust
            n_ptr += 1
            i += 1
            prev_const = False
            if i < len(toks) and toks[i] == "const":
                prev_const = True
                i += 1
        if n_ptr == 0 and base == "void":
            return "" if returning else "c_void"
        return rust

    def _param(self, a: str, index: int):
        m = re.fullmatch(r"(.*?)(\w+)", a)
        if not m or m.group(2) in self.known or m.group(2) == "const":   # no name (or only a type)
            return (f"arg{index}", self._type(a))
        name = m.group(2)
        if name in ("type", "match", "fn", "in", "ref", "loop", "move", "box", "impl"):
            name += "_"
        return (name, self._type(m.group(1).strip()))

    def rust(self) -> str:
        out = ["// Generated by tools/gen_rust_ffi.py from include/spf_hip.h — do not edit; `make -C spf_amd/csrc rust-ffi` regenerates it.",
               "// The documentation of every item is in the header.",
               "#![allow(non_camel_case_types, dead_code)]",
               "use std::os::raw::{c_char, c_int, c_void};", "",
               "pub type spf_status = c_int;"]
        for name, consts in self.enums:
            if name:
                out.append(f"pub type {name} = c_int;")
            ty = name or "spf_status"
            for cname, val in consts:
                out.append(f"pub const {cname}: {ty} = {val};")
        out.append("")
        for name in self.opaque:
            if any(name == s for s, _ in self.structs):
                continue
            out += ["#[repr(C)]", f"pub struct {name} {{ _private: [u8; 0] }}"]
        for name, fields in self.structs:
            out += ["#[repr(C)]", "#[derive(Clone, Copy, Debug, Default)]", f"pub struct {name} {{"]
            out += [f"    pub {f}: {t}," for f, t in fields]
            out.append("}")
        out += ["", '#[link(name = "spf_hip")]', 'extern "C" {']
        for name, args, ret in self.functions:
            sig = ", ".join(f"{a}: {t}" for a, t in args)
            out.append(f"    pub fn {name}({sig})" + (f" -> {ret};" if ret else ";"))
        out.append("}")
        return "\n".join(out) + "\n"


def exported_symbols(lib: str):
    txt = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in txt.splitlines() if " T " in l and l.split()[-1].startswith("spf_"))


FAMILIES = [
    ("context and keys", r"spf_(default_params|create|destroy|last_error|version|load_|key_blob)"),
    ("hot path, host pointers (`_batch`)", r"spf_(?!group_|pool_).*_batch$"),
    ("hot path, device pointers (`_dev`) and device buffers", r"spf_((?!group_).*_dev$|device_)"),
    ("call coalescing: the pool, host pointers", r"spf_pool_(?!.*_v$)(?!value_|trim|flush|create_group)"),
    ("device-resident values and the pool by handle", r"spf_(value_|pool_.*_v$|pool_value_stats|pool_trim|pool_flush)"),
    ("gate graphs", r"spf_graph_"),
    ("device group (every GPU, one process)", r"spf_(group_|pool_create_group)"),
    ("measurement, LUT, wire formats, constants", r"spf_"),
]


def index_markdown(h: "Header", header_path: str) -> str:
    """every function of the header by family, with the line it is declared on"""
    lines = open(header_path).read().splitlines()
    where = {}
    for name, _, _ in h.functions:
        for i, l in enumerate(lines, 1):
            if re.search(rf"\b{name}\s*\(", l) and not l.lstrip().startswith(("*", "/*")):
                where[name] = i
                break
    left = [n for n, _, _ in h.functions]
    out = []
    for title, pat in FAMILIES:
        mine = [n for n in left if re.match(pat, n)]
        left = [n for n in left if n not in mine]
        if mine:
            out.append(f"- **{title}** ({len(mine)}): " + ", ".join(f"`{n}` (:{where.get(n, 0)})" for n in mine))
    return "\n".join(out) + "\n"


def main(argv):
    if len(argv) == 2 and argv[0] == "--index":
        sys.stdout.write(index_markdown(Header(argv[1]), argv[1]))
        return 0
    if len(argv) >= 3 and argv[0] == "--check-library":
        h = Header(argv[2])
        want = sorted(n for n, _, _ in h.functions)
        have = exported_symbols(argv[1])
        if want != have:
            print("header only:", sorted(set(want) - set(have)), "\nlibrary only:", sorted(set(have) - set(want)), file=sys.stderr)
            return 1
        print(f"{len(want)} functions: the library exports exactly what the header declares")
        return 0
    if len(argv) != 1:
        print(__doc__, file=sys.stderr)
        return 2
    sys.stdout.write(Header(argv[0]).rust())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
