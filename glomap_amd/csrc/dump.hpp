// dump.hpp — flat on-disk problem format.
//
// With a dump directory set (gsfm_ctx_set_dump_dir, or GSFM_DUMP_DIR in the environment when the ctx is created)
// every gsfm_{ra,gp,ba}_solve writes `<dir>/<kind>_<seq>.gsfm`: the flat structure-of-arrays problem exactly as it
// crossed the C ABI, the options, and the results + report of the solve.  A GLOMAP build linked against libgsfm thus
// turns a real dataset into replayable files at the three call sites (global_mapper.cc:92,160,209) without any change
// to GLOMAP; `tools/replay.py` re-runs them through the C ABI on a box that has neither GLOMAP nor COLMAP.
//
// Layout: 8 bytes magic "GSFMFLT1", uint64 little-endian header length H, H bytes of JSON
//   {"kind": "ra"|"gp"|"ba", "scalars": {...}, "options": {...}, "report": {...}, "status": rc,
//    "arrays": [{"name", "dtype": "f64"|"i32"|"i64"|"u8"|"u32", "shape": [...], "offset", "nbytes"}, ...]}
// then the raw little-endian array data; offsets are relative to the end of the header, 64-byte aligned.
#pragma once

#include <cinttypes>
#include <sstream>
#include <type_traits>

#include "common.hpp"

namespace gsfm {

class FlatDump {
 public:
  FlatDump(gsfm_ctx* ctx, const char* kind) : ctx_(ctx), kind_(kind) {}
  bool active() const { return ctx_ != nullptr && !ctx_->dump_dir.empty(); }

  template <typename T>
  void array(const char* name, const T* ptr, std::vector<int64_t> shape, int mem) {
    if (!active() || ptr == nullptr) return;
    size_t n = 1;
    for (int64_t s : shape) n *= (size_t)s;
    Arr a;
    a.name = name;
    a.dtype = dtype_of<T>();
    a.shape = std::move(shape);
    a.data.resize(n * sizeof(T));
    if (n > 0) {
      if (mem == GSFM_MEM_DEVICE) {
        if (hipMemcpy(a.data.data(), ptr, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return;
      } else {
        std::memcpy(a.data.data(), ptr, n * sizeof(T));
      }
    }
    arrays_.push_back(std::move(a));
  }
  void scalar(const char* name, double v) { field(scalars_, name, v); }
  void option(const char* name, double v) { field(options_, name, v); }

  void write(const gsfm_report* rep, int status) {
    if (!active()) return;
    std::ostringstream js;
    js.precision(17);
    js << "{\"format\": \"gsfm-flat\", \"version\": 1, \"kind\": \"" << kind_ << "\", \"status\": " << status << ", \"scalars\": {"
       << scalars_.str() << "}, \"options\": {" << options_.str() << "}, \"report\": {";
    if (rep)
      js << "\"iterations\": " << rep->iterations << ", \"iterations_l1\": " << rep->iterations_l1
         << ", \"iterations_irls\": " << rep->iterations_irls << ", \"successful_steps\": " << rep->successful_steps
         << ", \"linear_iterations\": " << rep->linear_iterations << ", \"initial_cost\": " << num(rep->initial_cost)
         << ", \"final_cost\": " << num(rep->final_cost) << ", \"termination\": " << rep->termination
         << ", \"seconds_total\": " << num(rep->seconds_total);
    js << "}, \"arrays\": [";
    size_t off = 0;
    for (size_t i = 0; i < arrays_.size(); ++i) {
      const Arr& a = arrays_[i];
      js << (i ? ", " : "") << "{\"name\": \"" << a.name << "\", \"dtype\": \"" << a.dtype << "\", \"shape\": [";
      for (size_t k = 0; k < a.shape.size(); ++k) js << (k ? ", " : "") << a.shape[k];
      js << "], \"offset\": " << off << ", \"nbytes\": " << a.data.size() << "}";
      off = (off + a.data.size() + 63) / 64 * 64;
    }
    js << "]}";
    std::string header = js.str();
    while ((16 + header.size()) % 64) header.push_back(' ');
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s_%04d.gsfm", ctx_->dump_dir.c_str(), kind_, ctx_->dump_seq++);
    FILE* f = fopen(path, "wb");
    if (!f) {
      fprintf(stderr, "[gsfm] cannot write %s\n", path);
      return;
    }
    const uint64_t hl = header.size();
    fwrite("GSFMFLT1", 1, 8, f);
    fwrite(&hl, sizeof(hl), 1, f);
    fwrite(header.data(), 1, header.size(), f);
    static const char zeros[64] = {0};
    for (const Arr& a : arrays_) {
      fwrite(a.data.data(), 1, a.data.size(), f);
      const size_t pad = (64 - a.data.size() % 64) % 64;
      fwrite(zeros, 1, pad, f);
    }
    fclose(f);
  }

 private:
  struct Arr {
    std::string name;
    const char* dtype;
    std::vector<int64_t> shape;
    std::vector<unsigned char> data;
  };
  template <typename T>
  static const char* dtype_of() {
    if (std::is_same<T, double>::value) return "f64";
    if (std::is_same<T, int32_t>::value) return "i32";
    if (std::is_same<T, int64_t>::value || std::is_same<T, long>::value) return "i64";
    if (std::is_same<T, uint8_t>::value) return "u8";
    return "u32";
  }
  static std::string num(double v) {
    if (!std::isfinite(v)) return "null";
    char buf[64];
    snprintf(buf, sizeof(buf), "%.17g", v);
    return buf;
  }
  static void field(std::ostringstream& os, const char* name, double v) {
    if (os.tellp() > 0) os << ", ";
    os << "\"" << name << "\": " << num(v);
  }
  gsfm_ctx* ctx_;
  const char* kind_;
  std::vector<Arr> arrays_;
  std::ostringstream scalars_, options_;
};

#define GSFM_DUMP_OPT(dump, o, f) (dump).option(#f, (double)(o)->f)

inline void dump_lm_options(FlatDump& d, const gsfm_lm_options* lm) {
  GSFM_DUMP_OPT(d, lm, max_num_iterations);
  GSFM_DUMP_OPT(d, lm, function_tolerance);
  GSFM_DUMP_OPT(d, lm, gradient_tolerance);
  GSFM_DUMP_OPT(d, lm, parameter_tolerance);
  GSFM_DUMP_OPT(d, lm, initial_trust_region_radius);
  GSFM_DUMP_OPT(d, lm, max_trust_region_radius);
  GSFM_DUMP_OPT(d, lm, min_trust_region_radius);
  GSFM_DUMP_OPT(d, lm, min_relative_decrease);
  GSFM_DUMP_OPT(d, lm, min_lm_diagonal);
  GSFM_DUMP_OPT(d, lm, max_lm_diagonal);
  GSFM_DUMP_OPT(d, lm, jacobi_scaling);
  GSFM_DUMP_OPT(d, lm, max_num_consecutive_invalid_steps);
  GSFM_DUMP_OPT(d, lm, pcg_relative_tolerance);
  GSFM_DUMP_OPT(d, lm, pcg_max_iterations);
  GSFM_DUMP_OPT(d, lm, max_num_line_search_step_size_iterations);
}

}  // namespace gsfm
