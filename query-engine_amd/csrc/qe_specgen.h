// Text of the compile-time description ("Spec") of one lowered launch: the struct that
// qe_static.h's kernel templates take.  One emitter serves both users, so they cannot drift apart:
// the build-time generator (through qe_lowering_describe) and the run-time compiler (qe_jit.cpp).
#pragma once
#include <sstream>
#include <string>
#include <vector>

#include "qe_signature.h"

namespace qe {

struct SpecFields {
  const VmProgram *prog = nullptr;
  const VmOut *outs = nullptr;
  int n_outs = 0;
  uint32_t pred_pc = 0, pred_cache_mask = 0, out_cache_mask = 0;
  const VmTask *tasks = nullptr;
  int n_tasks = 0;
  const VmKey *keys = nullptr;
  int n_keys = 0;
  const VmAcc *accs = nullptr;
  int n_accs = 0;
  std::vector<uint32_t> sig;
};

inline void spec_array(std::ostringstream &o, const char *name, const std::vector<uint32_t> &v, size_t size) {
  std::vector<uint32_t> w = v;
  if (w.size() < size) w.resize(size, 0);
  if (w.empty()) w.push_back(0);
  o << "  static constexpr uint32_t " << name << "[" << w.size() << "] = {";
  for (size_t i = 0; i < w.size(); ++i) o << (i ? ", " : "") << w[i] << "u";
  o << "};\n";
}

// `struct_name`: C++ identifier; `label`: the kernel's display name in profiles.
inline std::string spec_struct_source(const std::string &struct_name, const std::string &label, const SpecFields &f) {
  const VmProgram &P = *f.prog;
  std::ostringstream o;
  o << "struct " << struct_name << " {\n";
  o << "  static constexpr const char *name = \"" << label << "\";\n";
  int n_code = program_words_used(P);
  o << "  static constexpr int n_code = " << n_code << ";\n";
  spec_array(o, "code", std::vector<uint32_t>(P.insn, P.insn + n_code), 0);
  o << "  static constexpr int cache_n = " << P.cache_n << ";\n";
  std::vector<uint32_t> kind, nullable;
  for (uint32_t i = 0; i < P.cache_n; ++i) {
    kind.push_back(P.cache_kind[i]);
    nullable.push_back(P.cols[P.cache_col[i]].validity ? 1u : 0u);
  }
  spec_array(o, "cache_kind", kind, VM_NCACHE);
  spec_array(o, "cache_nullable", nullable, VM_NCACHE);
  o << "  static constexpr int n_outs = " << f.n_outs << ";\n";
  std::vector<uint32_t> a, b, c, d;
  for (int i = 0; i < f.n_outs; ++i) {
    a.push_back(f.outs[i].pc);
    b.push_back(f.outs[i].type);
    c.push_back(f.outs[i].validity ? 1u : 0u);
  }
  spec_array(o, "out_pc", a, VM_MAX_OUTS);
  spec_array(o, "out_type", b, VM_MAX_OUTS);
  spec_array(o, "out_nullable", c, VM_MAX_OUTS);
  o << "  static constexpr uint32_t pred_pc = " << f.pred_pc << "u, pred_cache_mask = " << f.pred_cache_mask
    << "u, out_cache_mask = " << f.out_cache_mask << "u;\n";
  o << "  static constexpr int n_tasks = " << f.n_tasks << ";\n";
  a.clear(), b.clear(), c.clear(), d.clear();
  for (int i = 0; i < f.n_tasks; ++i) {
    a.push_back(f.tasks[i].pc);
    b.push_back(f.tasks[i].role);
    c.push_back(f.tasks[i].idx);
    d.push_back(f.tasks[i].has_eval);
  }
  spec_array(o, "task_pc", a, VM_MAX_TASKS);
  spec_array(o, "task_role", b, VM_MAX_TASKS);
  spec_array(o, "task_idx", c, VM_MAX_TASKS);
  spec_array(o, "task_eval", d, VM_MAX_TASKS);
  o << "  static constexpr int n_keys = " << f.n_keys << ";\n";
  a.clear(), b.clear(), c.clear();
  for (int i = 0; i < f.n_keys; ++i) {
    a.push_back(f.keys[i].shift);
    b.push_back(f.keys[i].bits);
  }
  spec_array(o, "key_shift", a, VM_MAX_KEYS);
  spec_array(o, "key_bits", b, VM_MAX_KEYS);
  o << "  static constexpr int n_accs = " << f.n_accs << ";\n";
  a.clear(), b.clear(), c.clear();
  for (int i = 0; i < f.n_accs; ++i) {
    a.push_back(f.accs[i].op);
    b.push_back(f.accs[i].in_type);
    c.push_back(f.accs[i].track_nn);
  }
  spec_array(o, "acc_op", a, VM_MAX_ACCS);
  spec_array(o, "acc_type", b, VM_MAX_ACCS);
  spec_array(o, "acc_track", c, VM_MAX_ACCS);
  o << "  static constexpr int n_sig = " << f.sig.size() << ";\n";
  spec_array(o, "sig", f.sig, 0);
  o << "};\n";
  return o.str();
}

inline SpecFields spec_fields(const ProjectArgs &A) {
  SpecFields f;
  f.prog = &A.prog;
  f.outs = A.outs;
  f.n_outs = A.n_outs;
  f.sig = signature(A);
  return f;
}
inline SpecFields spec_fields(const FilterArgs &A) {
  SpecFields f;
  f.prog = &A.prog;
  f.outs = A.outs;
  f.n_outs = A.n_outs;
  f.pred_pc = A.pred_pc;
  f.pred_cache_mask = A.pred_cache_mask;
  f.out_cache_mask = A.out_cache_mask;
  f.sig = signature(A);
  return f;
}
inline SpecFields spec_fields(const AggArgs &A) {
  SpecFields f;
  f.prog = &A.prog;
  f.tasks = A.tasks;
  f.n_tasks = A.n_tasks;
  f.keys = A.keys;
  f.n_keys = A.n_keys;
  f.accs = A.accs;
  f.n_accs = A.n_accs;
  f.sig = signature(A);
  return f;
}

// The same shape as the first pass of the high-cardinality mode: a kernel of its own, an entry of its own.
inline SpecFields spec_fields_spill(const AggArgs &A) {
  SpecFields f = spec_fields(A);
  f.sig.push_back(0x5104u);
  return f;
}
// ... and as the first pass that partitions its rows itself.
inline SpecFields spec_fields_direct(const AggArgs &A) {
  SpecFields f = spec_fields(A);
  f.sig.push_back(0xD12Eu);
  return f;
}

}  // namespace qe
