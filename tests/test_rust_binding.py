"""The Rust side shipped as text (rust/qe-gpu-sys, rust/gpu_exec.rs; no Rust toolchain where it was written): the raw
declarations cover include/qe_gpu.h symbol for symbol with the same number of arguments and are the generator's
current output; the binding over the reference's traits (PhysicalExpression, Accumulator, PhysicalAggregateExpression:
physical_expressions.rs:22-24, 351-362; PhysicalPlan: physical_plan/mod.rs:34-66) calls only functions that exist, with
the declared number of arguments."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_RS = os.path.join(ROOT, "rust", "qe-gpu-sys", "src", "lib.rs")


def _header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qe_gpu.h")).read(), flags=re.S)
    text = re.sub(r"#.*", "", text)
    out = {}
    for m in re.finditer(r"\b(qe_\w+)\s*\(([^()]*)\)\s*;", text):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (\w+)\(([^)]*)\)", open(LIB_RS).read()):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else args.count(",") + 1
    return out


def test_every_header_symbol_is_declared_with_the_same_arity():
    h, r = _header_functions(), _rust_functions()
    assert len(h) > 70
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert {n: a for n, a in h.items() if r[n] != a} == {}


def test_the_library_exports_what_the_crate_declares():
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "query-engine_amd", "libqe_gpu.so"))
    for name in _rust_functions():
        assert hasattr(lib, name), name


def test_generated_file_is_current():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])


def test_binding_calls_only_declared_functions_with_their_arity():
    src = open(os.path.join(ROOT, "rust", "gpu_exec.rs")).read()
    declared = _rust_functions()
    calls = list(re.finditer(r"sys::(qe_\w+)\(", src))
    assert len(calls) > 30
    for m in calls:
        name = m.group(1)
        assert name in declared, name
        # count top-level commas of the call's argument list
        depth, i, commas, empty = 1, m.end(), 0, True
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                commas += 1
            if depth and not ch.isspace():
                empty = False
            i += 1
        n_args = 0 if empty else commas + 1
        assert n_args == declared[name], (name, n_args, declared[name])
    # the traits of the reference it implements
    for needle in ("impl PhysicalExpression for GpuExpr", "impl Accumulator for GpuAccumulator",
                   "impl PhysicalAggregateExpression for GpuAggregate", "fn evaluate(&self, input: &Chunk<Arc<dyn Array>>) -> Result<ColumnarValue, Error>",
                   "fn accumulate(&mut self, input: &Vec<ColumnarValue>, validity: Option<&Bitmap>) -> Result<(), Error>",
                   "fn final_value(self: Box<Self>) -> Result<ColumnarValue, Error>", "fn create_accumulator(&self, _index: usize) -> Box<dyn Accumulator>",
                   "pub fn execute(self) -> Result<Box<dyn Iterator<Item = Batch>>, Error>"):
        assert needle in src, needle
    # constants it names exist in the generated crate
    consts = set(re.findall(r"pub const (\w+):", open(LIB_RS).read()))
    for name in set(re.findall(r"sys::(QE_\w+)", src)):
        assert name in consts, name
