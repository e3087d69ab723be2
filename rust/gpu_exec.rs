// NOT COMPILED where it was written (no Rust toolchain in that image).  The file a maintainer of JanKaul/query-engine
// would add as `query-engine/src/physical_plan/gpu.rs` (plus `pub mod gpu;` in physical_plan/mod.rs, a
// `Gpu(gpu::GpuExec)` variant in `enum PhysicalPlan` and `qe-gpu-sys` in Cargo.toml): the reference's own traits
// implemented over the C ABI of libqe_gpu.so.
//
//   impl PhysicalExpression for GpuExpr            physical_expressions.rs:22-24   -> qe_expr_evaluate / qe_value
//   impl Accumulator for GpuAccumulator            physical_expressions.rs:351-358 -> qe_agg_update / qe_agg_finalize
//   impl PhysicalAggregateExpression for GpuAggregate  :360-362                    -> qe_agg_state_new
//   GpuExec + PhysicalPlan::Gpu                    physical_plan/mod.rs:34-66      -> qe_plan_* / qe_plan_execute / qe_iter_next
//
// Arrays cross the boundary as raw Arrow buffers (qe_column: validity, offsets, values - the order of the Arrow C data
// interface); nothing is copied on the way in (qe_batch_import borrows), one copy per direction over the host link
// (qe_batch_to_device / qe_batch_to_host).  Errors come back as the reference's own `Error` variants: status codes
// 10..24 are its enum in declaration order (error.rs:3-35).
use std::ffi::CStr;
use std::fmt::{self, Display};
use std::os::raw::c_void;
use std::ptr;
use std::sync::Arc;

use arrow2::array::{Array, BooleanArray, PrimitiveArray, Utf8Array};
use arrow2::bitmap::Bitmap;
use arrow2::buffer::Buffer;
use arrow2::chunk::Chunk;
use arrow2::datatypes::{DataType, PhysicalType, PrimitiveType, Schema};
use arrow2::scalar::{BooleanScalar, NullScalar, PrimitiveScalar, Scalar, Utf8Scalar};

use qe_gpu_sys as sys;

use crate::columnar_value::ColumnarValue;
use crate::error::Error;
use crate::physical_plan::physical_expressions::{Accumulator, PhysicalAggregateExpression, PhysicalExpression};
use crate::physical_plan::{Batch, PhysicalPlan};

// ------------------------------------------------------------------ errors
fn last_message() -> String {
    unsafe { CStr::from_ptr(sys::qe_last_error_message()) }.to_string_lossy().into_owned()
}

/// qe_status -> the reference's Error (error.rs:3-35; the codes follow the enum's declaration order).
fn check(status: i32) -> Result<(), Error> {
    let msg = || last_message();
    Err(match status {
        sys::QE_OK => return Ok(()),
        sys::QE_ERR_EXCEEDING_BOUNDS => Error::ExceedingBoundsError(0),
        sys::QE_ERR_NO_FIELD_IN_LOGICAL_PLAN => Error::NoFieldInLogicalPlan(msg()),
        sys::QE_ERR_DIFFERENT_SIZES => Error::DifferentSizes(msg(), String::new()),
        sys::QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED => Error::PhysicalExpressionNotSuported(msg()),
        sys::QE_ERR_PHYSICAL_PLAN_NOT_SUPPORTED => Error::PhysicalPlanNotSuported(msg()),
        sys::QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED => Error::PhysicalTypeNotSuported(msg()),
        sys::QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED => Error::PrimitiveTypeNotSuported(msg()),
        sys::QE_ERR_MISSING_CHILDREN => Error::MissingChildren(msg()),
        sys::QE_ERR_MISSING_INPUT_PHYSICAL_PLAN => Error::MissingInputPhysicalPlan(msg()),
        sys::QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE => Error::EmptyHashmapForAggregate,
        sys::QE_ERR_DOWNCAST => Error::DowncastError,
        sys::QE_ERR_SCALAR_TO_ARRAY => Error::ScalarToArrayError(msg()),
        sys::QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER => Error::NoBooleanArrayForFilter,
        sys::QE_ERR_IO => Error::IoError(std::io::Error::new(std::io::ErrorKind::Other, msg())),
        _ => Error::ArrowError(arrow2::error::ArrowError::ExternalFormat(msg())), // 24 and the backend-only codes 40..44
    })
}

// ----------------------------------------------------------------- context
/// One HIP device + stream.  Not Send / Sync, like everything in the reference (physical_plan/mod.rs:58).
pub struct GpuContext(*mut sys::qe_context);

impl GpuContext {
    pub fn new(device: i32) -> Result<Arc<Self>, Error> {
        let mut h = ptr::null_mut();
        check(unsafe { sys::qe_context_create(device, ptr::null_mut(), &mut h) })?;
        Ok(Arc::new(GpuContext(h)))
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { sys::qe_context_free(self.0) }
    }
}

// ------------------------------------------------------ chunks <-> batches
struct OwnedBatch(*mut sys::qe_batch);
impl Drop for OwnedBatch {
    fn drop(&mut self) {
        unsafe { sys::qe_batch_free(self.0) }
    }
}

fn describe(array: &dyn Array) -> Result<sys::qe_column, Error> {
    let validity = array.validity().map(|b| b.as_slice()); // (bytes, bit offset, length); chunks off the Parquet reader have offset 0
    let null_ptr = validity.map(|(bytes, offset, _)| {
        assert_eq!(offset, 0, "sliced bitmaps are re-packed by the caller");
        bytes.as_ptr() as *const c_void
    });
    let mut c = sys::qe_column {
        type_: 0,
        memory_space: sys::QE_MEM_HOST,
        length: array.len() as i64,
        null_count: array.null_count() as i64,
        validity: null_ptr.unwrap_or(ptr::null()),
        offsets: ptr::null(),
        values: ptr::null(),
        values_bytes: 0,
    };
    match array.data_type().to_physical_type() {
        PhysicalType::Primitive(PrimitiveType::Int32) => {
            let a = array.as_any().downcast_ref::<PrimitiveArray<i32>>().ok_or(Error::DowncastError)?;
            c.type_ = sys::QE_TYPE_INT32;
            c.values = a.values().as_ptr() as *const c_void;
        }
        PhysicalType::Primitive(PrimitiveType::Int64) => {
            let a = array.as_any().downcast_ref::<PrimitiveArray<i64>>().ok_or(Error::DowncastError)?;
            c.type_ = if matches!(array.data_type(), DataType::Timestamp(_, _)) { sys::QE_TYPE_TIMESTAMP_NS } else { sys::QE_TYPE_INT64 };
            c.values = a.values().as_ptr() as *const c_void;
        }
        PhysicalType::Primitive(PrimitiveType::Float64) => {
            let a = array.as_any().downcast_ref::<PrimitiveArray<f64>>().ok_or(Error::DowncastError)?;
            c.type_ = sys::QE_TYPE_FLOAT64;
            c.values = a.values().as_ptr() as *const c_void;
        }
        PhysicalType::Boolean => {
            let a = array.as_any().downcast_ref::<BooleanArray>().ok_or(Error::DowncastError)?;
            let (bytes, offset, _) = a.values().as_slice();
            assert_eq!(offset, 0);
            c.type_ = sys::QE_TYPE_BOOLEAN;
            c.values = bytes.as_ptr() as *const c_void;
        }
        PhysicalType::Utf8 => {
            let a = array.as_any().downcast_ref::<Utf8Array<i32>>().ok_or(Error::DowncastError)?;
            c.type_ = sys::QE_TYPE_UTF8;
            c.offsets = a.offsets().as_ptr() as *const c_void;
            c.values = a.values().as_ptr() as *const c_void;
            c.values_bytes = a.values().len() as i64;
        }
        other => return Err(Error::PhysicalTypeNotSuported(format!("{:?}", other))),
    }
    Ok(c)
}

/// Chunk -> device-resident batch.  The import borrows the chunk's buffers only for the duration of the upload.
fn upload(ctx: &GpuContext, chunk: &Chunk<Arc<dyn Array>>) -> Result<OwnedBatch, Error> {
    let cols = chunk.arrays().iter().map(|a| describe(a.as_ref())).collect::<Result<Vec<_>, _>>()?;
    let mut host = ptr::null_mut();
    check(unsafe { sys::qe_batch_import(ctx.0, cols.as_ptr(), cols.len() as i32, &mut host) })?;
    let host = OwnedBatch(host);
    let mut dev = ptr::null_mut();
    check(unsafe { sys::qe_batch_to_device(ctx.0, host.0, &mut dev) })?;
    Ok(OwnedBatch(dev))
}

unsafe fn copy_out<T: Copy>(p: *const c_void, n: usize) -> Vec<T> {
    std::slice::from_raw_parts(p as *const T, n).to_vec()
}

/// Device batch -> Chunk of arrow2 arrays (one copy over the host link, then one into arrow2-owned buffers; handing
/// the library's buffers to arrow2 as foreign memory instead would save the second).
fn download(ctx: &GpuContext, batch: &OwnedBatch) -> Result<Chunk<Arc<dyn Array>>, Error> {
    let mut host = ptr::null_mut();
    check(unsafe { sys::qe_batch_to_host(ctx.0, batch.0, &mut host) })?;
    let host = OwnedBatch(host);
    let mut n_cols = 0i32;
    check(unsafe { sys::qe_batch_num_columns(host.0, &mut n_cols) })?;
    let mut arrays: Vec<Arc<dyn Array>> = Vec::with_capacity(n_cols as usize);
    for i in 0..n_cols {
        let mut c: sys::qe_column = unsafe { std::mem::zeroed() };
        check(unsafe { sys::qe_batch_column(host.0, i, &mut c) })?;
        let n = c.length as usize;
        let validity = if c.validity.is_null() {
            None
        } else {
            Some(Bitmap::from_u8_vec(unsafe { copy_out::<u8>(c.validity, (n + 7) / 8) }, n))
        };
        let array: Arc<dyn Array> = match c.type_ {
            sys::QE_TYPE_INT32 => Arc::new(PrimitiveArray::<i32>::from_data(DataType::Int32, Buffer::from(unsafe { copy_out::<i32>(c.values, n) }), validity)),
            sys::QE_TYPE_INT64 => Arc::new(PrimitiveArray::<i64>::from_data(DataType::Int64, Buffer::from(unsafe { copy_out::<i64>(c.values, n) }), validity)),
            sys::QE_TYPE_TIMESTAMP_NS => Arc::new(PrimitiveArray::<i64>::from_data(
                DataType::Timestamp(arrow2::datatypes::TimeUnit::Nanosecond, None),
                Buffer::from(unsafe { copy_out::<i64>(c.values, n) }),
                validity,
            )),
            sys::QE_TYPE_FLOAT64 => Arc::new(PrimitiveArray::<f64>::from_data(DataType::Float64, Buffer::from(unsafe { copy_out::<f64>(c.values, n) }), validity)),
            sys::QE_TYPE_BOOLEAN => Arc::new(BooleanArray::from_data(DataType::Boolean, Bitmap::from_u8_vec(unsafe { copy_out::<u8>(c.values, (n + 7) / 8) }, n), validity)),
            sys::QE_TYPE_UTF8 => Arc::new(Utf8Array::<i32>::from_data(
                DataType::Utf8,
                Buffer::from(unsafe { copy_out::<i32>(c.offsets, n + 1) }),
                Buffer::from(unsafe { copy_out::<u8>(c.values, c.values_bytes as usize) }),
                validity,
            )),
            _ => return Err(Error::DowncastError),
        };
        arrays.push(array);
    }
    Ok(Chunk::new(arrays))
}

// -------------------------------------------------------------- expressions
/// The operators of qe_expr_binary.  Eq / Neq and Add..Div are the reference's (physical_expressions.rs:232-233,
/// 346-349); the others exist only logically there (logical_expression.rs:284-294,342) and lower here.
#[derive(Clone, Copy, Debug)]
pub enum BinaryOp { Eq = 1, Neq, Gt, GtEq, Lt, LtEq, And, Or, Add, Sub, Mul, Div, Mod }

/// An expression tree described by value; the library handle is built per call (qe_expr_* consume their operands, the
/// way the reference's expression structs own theirs).
#[derive(Clone, Debug)]
pub enum GpuExprNode {
    Column(usize),
    LitInt(i32),
    LitInt64(i64),
    LitFloat(f64),
    LitBool(bool),
    LitString(String),
    Binary(BinaryOp, Box<GpuExprNode>, Box<GpuExprNode>),
}

impl GpuExprNode {
    unsafe fn build(&self) -> Result<*mut sys::qe_expr, Error> {
        let mut h = ptr::null_mut();
        match self {
            GpuExprNode::Column(i) => check(sys::qe_expr_column(*i as i32, &mut h))?,
            GpuExprNode::LitInt(v) => check(sys::qe_expr_lit_i32(*v, &mut h))?,
            GpuExprNode::LitInt64(v) => check(sys::qe_expr_lit_i64(*v, &mut h))?,
            GpuExprNode::LitFloat(v) => check(sys::qe_expr_lit_f64(*v, &mut h))?,
            GpuExprNode::LitBool(v) => check(sys::qe_expr_lit_bool(*v as i32, &mut h))?,
            GpuExprNode::LitString(s) => check(sys::qe_expr_lit_utf8(s.as_ptr() as *const _, s.len() as i64, &mut h))?,
            GpuExprNode::Binary(op, l, r) => {
                let lh = l.build()?;
                let rh = match r.build() {
                    Ok(x) => x,
                    Err(e) => {
                        sys::qe_expr_free(lh);
                        return Err(e);
                    }
                };
                check(sys::qe_expr_binary(*op as i32, lh, rh, &mut h))?; // consumes lh and rh, also on failure
            }
        }
        Ok(h)
    }
}

impl Display for GpuExprNode {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        // the library renders the reference's Display strings (physical_expressions.rs:64-68,225-229,338-342)
        unsafe {
            let h = self.build().map_err(|_| fmt::Error)?;
            let mut buf = vec![0u8; 4096];
            let mut need = 0i64;
            let st = sys::qe_expr_to_string(h, buf.as_mut_ptr() as *mut _, buf.len() as i64, &mut need);
            sys::qe_expr_free(h);
            if st != sys::QE_OK {
                return Err(fmt::Error);
            }
            write!(f, "{}", CStr::from_ptr(buf.as_ptr() as *const _).to_string_lossy())
        }
    }
}

/// `impl PhysicalExpression`: the whole tree is one fused kernel launch; a tree of literals folds to a Scalar on the host,
/// exactly the (Scalar, Scalar) arms of the reference (physical_expressions.rs:204-208, 269-321).
pub struct GpuExpr {
    pub ctx: Arc<GpuContext>,
    pub node: GpuExprNode,
}

impl Display for GpuExpr {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        self.node.fmt(f)
    }
}

impl PhysicalExpression for GpuExpr {
    fn evaluate(&self, input: &Chunk<Arc<dyn Array>>) -> Result<ColumnarValue, Error> {
        let batch = upload(&self.ctx, input)?;
        unsafe {
            let expr = self.node.build()?;
            let mut v: sys::qe_value = std::mem::zeroed();
            let st = sys::qe_expr_evaluate(self.ctx.0, expr, batch.0, &mut v);
            sys::qe_expr_free(expr);
            check(st)?;
            let result = if v.is_scalar != 0 {
                let scalar: Box<dyn Scalar> = if v.scalar_valid == 0 {
                    Box::new(NullScalar::new())
                } else {
                    match v.type_ {
                        sys::QE_TYPE_INT32 => Box::new(PrimitiveScalar::<i32>::new(DataType::Int32, Some(v.scalar_i64 as i32))),
                        sys::QE_TYPE_INT64 => Box::new(PrimitiveScalar::<i64>::new(DataType::Int64, Some(v.scalar_i64))),
                        sys::QE_TYPE_FLOAT64 => Box::new(PrimitiveScalar::<f64>::new(DataType::Float64, Some(v.scalar_f64))),
                        sys::QE_TYPE_BOOLEAN => Box::new(BooleanScalar::new(Some(v.scalar_i64 != 0))),
                        _ => {
                            let bytes = std::slice::from_raw_parts(v.scalar_utf8 as *const u8, v.scalar_utf8_len as usize);
                            Box::new(Utf8Scalar::<i32>::new(Some(String::from_utf8_lossy(bytes).into_owned())))
                        }
                    }
                };
                Ok(ColumnarValue::Scalar(scalar))
            } else {
                let one = OwnedBatch(v.array);
                v.array = ptr::null_mut();
                download(&self.ctx, &one).map(|chunk| ColumnarValue::Array(chunk.arrays()[0].clone()))
            };
            sys::qe_value_release(&mut v);
            result
        }
    }
}

// -------------------------------------------------------------- aggregates
#[derive(Clone, Copy, Debug)]
pub enum AggKind { Max = 1, Min, Sum, Count, Avg }

/// `impl PhysicalAggregateExpression`: Max / Min as in the reference (physical_expressions.rs:364-504) plus the Sum /
/// Count / Avg it declares only logically (logical_expression.rs:386-423).
pub struct GpuAggregate {
    pub ctx: Arc<GpuContext>,
    pub kind: AggKind,
    pub input: GpuExprNode,
    /// group expressions of the AggregateExec this aggregate belongs to: the accumulator's state is the device group
    /// table, which is keyed (the reference keeps one scalar per group and accumulator, physical_plan/mod.rs:343-363)
    pub group_exprs: Vec<GpuExprNode>,
}

impl Display for GpuAggregate {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "{:?}({})", self.kind, self.input)
    }
}

impl PhysicalExpression for GpuAggregate {
    // MaxExpression::evaluate delegates to its child (physical_expressions.rs:474-478)
    fn evaluate(&self, input: &Chunk<Arc<dyn Array>>) -> Result<ColumnarValue, Error> {
        GpuExpr { ctx: self.ctx.clone(), node: self.input.clone() }.evaluate(input)
    }
}

impl PhysicalAggregateExpression for GpuAggregate {
    fn create_accumulator(&self, _index: usize) -> Box<dyn Accumulator> {
        Box::new(GpuAccumulator { ctx: self.ctx.clone(), state: ptr::null_mut(), spec: (self.kind, self.input.clone(), self.group_exprs.clone()), kept: Vec::new() })
    }
}

/// `impl Accumulator` over qe_agg_state_new / qe_agg_update / qe_agg_finalize.  One accumulator object serves every
/// group: `accumulate` folds a whole batch into the device table in one launch (the reference calls its accumulators
/// once per group with a membership mask, physical_plan/mod.rs:326-364 - O(rows x groups)); the mask argument is
/// therefore expected to be None (AggregateExec's GPU arm passes the batch once).
pub struct GpuAccumulator {
    ctx: Arc<GpuContext>,
    state: *mut sys::qe_agg_state,
    spec: (AggKind, GpuExprNode, Vec<GpuExprNode>),
    kept: Vec<OwnedBatch>, // batches stay borrowed until the state is freed (a table that outgrows its hint folds them in again)
}

impl GpuAccumulator {
    unsafe fn ensure_state(&mut self) -> Result<(), Error> {
        if !self.state.is_null() {
            return Ok(());
        }
        let mut groups = Vec::new();
        for g in &self.spec.2 {
            groups.push(g.build()?);
        }
        let input = self.spec.1.build()?;
        let mut agg = ptr::null_mut();
        check(sys::qe_agg_expr_new(self.spec.0 as i32, input, &mut agg))?; // consumes input
        let aggs = [agg];
        check(sys::qe_agg_state_new(self.ctx.0, groups.as_ptr(), groups.len() as i32, aggs.as_ptr(), 1, sys::QE_AGG_MODE_SINGLE, 0, &mut self.state))
        // (consumes the expressions)
    }

    /// The whole output batch (key columns, then the aggregate): what AggregateExec assembles row by row
    /// (physical_plan/mod.rs:375-488).
    pub fn finish(mut self: Box<Self>) -> Result<Chunk<Arc<dyn Array>>, Error> {
        unsafe {
            self.ensure_state()?;
            let mut out = ptr::null_mut();
            check(sys::qe_agg_finalize(self.state, &mut out))?;
            download(&self.ctx, &OwnedBatch(out))
        }
    }
}

impl Accumulator for GpuAccumulator {
    fn accumulate(&mut self, input: &Vec<ColumnarValue>, validity: Option<&Bitmap>) -> Result<(), Error> {
        if validity.is_some() {
            return Err(Error::PhysicalExpressionNotSuported("per-group masks: the GPU accumulator groups on the device".to_string()));
        }
        let len = input.iter().find_map(|v| if let ColumnarValue::Array(a) = v { Some(a.len()) } else { None }).unwrap_or(1);
        let arrays: Vec<Arc<dyn Array>> = input
            .iter()
            .map(|v| match v {
                ColumnarValue::Array(a) => Ok(a.clone()),
                ColumnarValue::Scalar(s) => crate::columnar_value::scalar_to_array(s.clone(), len),
            })
            .collect::<Result<_, _>>()?;
        let batch = upload(&self.ctx, &Chunk::new(arrays))?;
        unsafe {
            self.ensure_state()?;
            check(sys::qe_agg_update(self.state, batch.0))?;
        }
        self.kept.push(batch);
        Ok(())
    }

    fn final_value(self: Box<Self>) -> Result<ColumnarValue, Error> {
        // one group (no group expressions): the single output cell as a scalar, like Max/MinAccumulator::final_value
        let chunk = self.finish()?;
        let col = chunk.arrays().last().ok_or(Error::EmptyHashmapForAggregate)?.clone();
        Ok(ColumnarValue::Array(col))
    }
}

impl Drop for GpuAccumulator {
    fn drop(&mut self) {
        if !self.state.is_null() {
            unsafe { sys::qe_agg_state_free(self.state) }
        }
    }
}

// -------------------------------------------------------------------- plan
/// The fifth PhysicalPlan variant: a whole Scan -> Selection -> Projection -> Aggregate chain handed to the library,
/// which fuses it into as few launches as it can and keeps every intermediate batch on the device.
pub enum GpuNode {
    /// chunks the CPU scan produced (ScanExec::execute, physical_plan/mod.rs:105-109), uploaded once
    ScanMemory(Vec<Chunk<Arc<dyn Array>>>),
    /// ScanExec over DataSource::Parquet, decoded on the GPU (data_source.rs:24-72)
    ScanParquet { path: String, projection: Option<Vec<String>> },
    Projection { input: Box<GpuNode>, exprs: Vec<GpuExprNode> },
    Selection { input: Box<GpuNode>, expr: GpuExprNode },
    Aggregate { input: Box<GpuNode>, group_exprs: Vec<GpuExprNode>, agg_exprs: Vec<(AggKind, Option<GpuExprNode>)> },
}

pub struct GpuExec {
    pub ctx: Arc<GpuContext>,
    pub root: GpuNode,
    pub schema: Schema,
}

impl GpuExec {
    pub fn schema(&self) -> &Schema {
        &self.schema
    }
    pub fn children(&self) -> Option<&[PhysicalPlan]> {
        None // the chain below lives inside the library
    }

    unsafe fn build(&self, node: &GpuNode, keep: &mut Vec<OwnedBatch>) -> Result<*mut sys::qe_plan, Error> {
        let mut plan = ptr::null_mut();
        match node {
            GpuNode::ScanMemory(chunks) => {
                let first = keep.len();
                for c in chunks {
                    keep.push(upload(&self.ctx, c)?);
                }
                let handles: Vec<*mut sys::qe_batch> = keep[first..].iter().map(|b| b.0).collect();
                check(sys::qe_plan_scan_memory(self.ctx.0, handles.as_ptr(), handles.len() as i32, &mut plan))?;
            }
            GpuNode::ScanParquet { path, projection } => {
                let cpath = std::ffi::CString::new(path.as_str()).map_err(|_| Error::DowncastError)?;
                let names: Vec<std::ffi::CString> = projection.iter().flatten().map(|s| std::ffi::CString::new(s.as_str()).unwrap()).collect();
                let ptrs: Vec<*const std::os::raw::c_char> = names.iter().map(|s| s.as_ptr()).collect();
                let n = if projection.is_some() { ptrs.len() as i32 } else { -1 };
                check(sys::qe_plan_scan_parquet(self.ctx.0, cpath.as_ptr(), ptrs.as_ptr(), n, &mut plan))?;
            }
            GpuNode::Projection { input, exprs } => {
                let child = self.build(input, keep)?;
                let hs = exprs.iter().map(|e| e.build()).collect::<Result<Vec<_>, _>>()?;
                check(sys::qe_plan_projection(child, hs.as_ptr(), hs.len() as i32, &mut plan))?; // consumes child and exprs
            }
            GpuNode::Selection { input, expr } => {
                let child = self.build(input, keep)?;
                check(sys::qe_plan_selection(child, expr.build()?, &mut plan))?;
            }
            GpuNode::Aggregate { input, group_exprs, agg_exprs } => {
                let child = self.build(input, keep)?;
                let gs = group_exprs.iter().map(|e| e.build()).collect::<Result<Vec<_>, _>>()?;
                let mut ags = Vec::new();
                for (kind, input) in agg_exprs {
                    let ih = match input {
                        Some(e) => e.build()?,
                        None => ptr::null_mut(), // COUNT(*)
                    };
                    let mut a = ptr::null_mut();
                    check(sys::qe_agg_expr_new(*kind as i32, ih, &mut a))?;
                    ags.push(a);
                }
                check(sys::qe_plan_aggregate(child, gs.as_ptr(), gs.len() as i32, ags.as_ptr(), ags.len() as i32, sys::QE_AGG_MODE_SINGLE, 0, &mut plan))?;
            }
        }
        Ok(plan)
    }

    /// PhysicalPlan::execute(self) (physical_plan/mod.rs:58-65): consumes the plan, yields batches.
    pub fn execute(self) -> Result<Box<dyn Iterator<Item = Batch>>, Error> {
        let mut keep = Vec::new();
        unsafe {
            let plan = self.build(&self.root, &mut keep)?;
            let mut it = ptr::null_mut();
            check(sys::qe_plan_execute(plan, &mut it))?; // consumes plan
            Ok(Box::new(GpuBatches { ctx: self.ctx.clone(), it, _inputs: keep }))
        }
    }
}

struct GpuBatches {
    ctx: Arc<GpuContext>,
    it: *mut sys::qe_iter,
    _inputs: Vec<OwnedBatch>, // scan batches are shared, not copied, by the plan: alive until the iterator is done
}

impl Iterator for GpuBatches {
    type Item = Batch; // Result<Chunk<Arc<dyn Array>>, Error>
    fn next(&mut self) -> Option<Batch> {
        let mut b = ptr::null_mut();
        match unsafe { sys::qe_iter_next(self.it, &mut b) } {
            sys::QE_END => None,
            sys::QE_OK => Some(download(&self.ctx, &OwnedBatch(b))),
            status => Some(check(status).map(|_| unreachable!())),
        }
    }
}

impl Drop for GpuBatches {
    fn drop(&mut self) {
        unsafe { sys::qe_iter_free(self.it) }
    }
}

// In physical_plan/mod.rs the enum gains `Gpu(gpu::GpuExec)` and its three methods one arm each:
//     PhysicalPlan::Gpu(g) => g.schema()      /      g.children()      /      g.execute()
// and query_planner.rs:105-166 builds a GpuNode chain instead of the CPU *Exec chain when a GpuContext is configured.
