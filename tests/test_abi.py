"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/qe_gpu.h declares, maps status codes onto the reference's Error variants, builds and prints
expression / plan trees, and refuses to run without a GPU (no CPU fallback).  No compute is called."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(qe):
    header = open(os.path.join(ROOT, "include", "qe_gpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(qe_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    raw = ctypes.CDLL(qe.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} is declared in qe_gpu.h but not exported"
    assert declared == set(qe.ABI), "python binding and header disagree"


def test_status_names_follow_reference_error_enum(qe):
    # error.rs:3-35, in declaration order, starting at 10
    variants = ["ExceedingBoundsError", "NoFieldInLogicalPlan", "DifferentSizes", "PhysicalExpressionNotSuported",
                "PhysicalPlanNotSuported", "PhysicalTypeNotSuported", "PrimitiveTypeNotSuported", "MissingChildren",
                "MissingInputPhysicalPlan", "EmptyHashmapForAggregate", "DowncastError", "ScalarToArrayError",
                "NoBooleanArrayForFilter", "IoError", "ArrowError"]
    for i, v in enumerate(variants):
        assert qe.lib.qe_status_name(10 + i).decode() == v
    assert qe.lib.qe_status_name(0).decode() == "Ok" and qe.lib.qe_status_name(1).decode() == "End"


def test_expression_display(qe):
    e = qe.EqExpression(qe.ColumnExpression(5), qe.LiteralStringExpression("Female"))
    assert str(e) == '#5 == #"Female"'
    e = qe.AddExpression(qe.ColumnExpression(0), qe.MulExpression(qe.ColumnExpression(1), qe.ColumnExpression(2)))
    assert str(e) == "#0 + #1 * #2"
    assert str(qe.LiteralIntegerExpression(-7)) == "#-7" and str(qe.LiteralBoolExpression(True)) == "#true"


def test_no_device_means_loud_failure(qe):
    if qe.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(qe.QeError) as e:
        qe.Context(0)
    assert e.value.name == "DeviceError" and "no CPU fallback" in str(e.value)


def test_invalid_arguments_do_not_crash(qe):
    h = ctypes.c_void_p()
    assert qe.lib.qe_expr_binary(99, None, None, ctypes.byref(h)) == 40  # InvalidArgument: null operands
    l, r = qe.ColumnExpression(0)._build(), qe.ColumnExpression(1)._build()
    assert qe.lib.qe_expr_binary(99, l, r, ctypes.byref(h)) == 13  # PhysicalExpressionNotSuported (consumes l, r)
    assert qe.lib.qe_agg_expr_new(3, None, ctypes.byref(h)) == 40  # only COUNT may omit its input
    assert qe.lib.qe_plan_selection(None, qe.ColumnExpression(0)._build(), ctypes.byref(h)) == 18  # MissingInputPhysicalPlan
    assert b"Selection" in qe.lib.qe_last_error_message()


def test_constants_match_the_header(qe):
    """Every QE_* enumerator the Python mirror carries has the header's value."""
    header = open(os.path.join(ROOT, "include", "qe_gpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    values = {name: int(v) for name, v in re.findall(r"\b(QE_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    assert len(values) >= 50
    mirror = {"QE_TYPE_": "TYPE_", "QE_OP_": "OP_", "QE_AGG_MODE_": "AGG_MODE_", "QE_AGG_": "AGG_", "QE_FLAG_": "FLAG_", "QE_MEM_": "MEM_"}
    checked = 0
    for name, value in values.items():
        for c_prefix, py_prefix in mirror.items():
            if name.startswith(c_prefix):
                py = py_prefix + name[len(c_prefix):]
                if hasattr(qe, py):
                    assert getattr(qe, py) == value, f"{py} != {name}"
                    checked += 1
                break
    assert checked >= 30
    assert qe.QE_OK == values["QE_OK"] and qe.QE_END == values["QE_END"]
