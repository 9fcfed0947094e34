// The library's configuration: ONE struct (include/sedef_hip.h: sdf_config), filled once -- from the SDF_* environment
// variables in sdf_create, or by the caller through sdf_create_cfg / sdf_config_set -- validated against the table below
// and kept by the context; nothing else in the library reads the environment for a setting.  (Round 4 had 47 getenv()
// calls spread over the planner, the launcher and the API, several behind function-local statics that froze the first
// value a process saw.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "sdf_ctx.h"

namespace sdf {

struct ConfigField {
  const char *env;   // the environment variable sdf_config_from_env reads; sdf_config_set takes it or the field name
  const char *name;  // field of sdf_config
  size_t offset;
  bool real;         // double (else int64_t)
  double def, lo, hi;
  const char *doc;
};
#define SDF_CF(envname, field, def, lo, hi, doc) {envname, #field, offsetof(sdf_config, field), false, (double)(def), (double)(lo), (double)(hi), doc}
static const ConfigField kConfigFields[] = {
    // ---- which kernel serves a task (tests force every kernel through these) ----
    SDF_CF("SDF_FORCE_GENERAL", force_general, 0, 0, 1, "1: every task on the LDS-resident general kernel"),
    SDF_CF("SDF_NO_PAIR", no_pair, 0, 0, 1, "1: never two tasks per wavefront (extz2_pair.hip off)"),
    SDF_CF("SDF_NO_MIXED", no_mixed, 0, 0, 1, "1: no mixed pairs (banded tasks of different lengths in one wavefront)"),
    SDF_CF("SDF_MIXED_MIN", mixed_min, 4096, 0, 1e12, "mixed-pair candidates a chunk must hold"),
    SDF_CF("SDF_SELF_PAIR_MAX", self_pair_max, 512, 0, 1e12, "tasks without a partner above which a chunk uses the one-task wave kernel"),
    SDF_CF("SDF_NO_STRIPE", no_stripe, 0, 0, 1, "1: no stripe kernels (wide full-band tasks stay on one workgroup)"),
    SDF_CF("SDF_STRIPE_MIN", stripe_min, 400, 128, 1e9, "full-band targets longer than this take the stripe kernel"),
    SDF_CF("SDF_STRIPE_NREG", stripe_nreg, 0, 0, 4, "stripe width in registers (1, 2, 4); 0: by the batch"),
    SDF_CF("SDF_STRIPE_CLAIM", stripe_claim, 1, 0, 1, "0: stripe workgroups take launch-order entry blockIdx.x instead of claiming one"),
    SDF_CF("SDF_STRIPE_SPIN_CAP", stripe_spin_cap, 1 << 24, 1, 2147483647.0, "polls before a stripe's wait gives its task up"),
    SDF_CF("SDF_BSTRIPE_MIN_ROWS", bstripe_min_rows, 4000, 0, 1e9, "banded tasks of this many anti-diagonals take the banded stripe kernel; 0: never"),
    SDF_CF("SDF_BSTRIPE_NREG", bstripe_nreg, 0, 0, 4, "banded stripe width in registers; 0: by the target length"),
    SDF_CF("SDF_BSTRIPE_ALL", bstripe_all, 0, 0, 1, "1: every long banded task on the banded stripe kernel"),
    SDF_CF("SDF_NO_STRIP", no_strip, 0, 0, 1, "1: no row-major strip kernels"),
    SDF_CF("SDF_STRIP_ALWAYS", strip_always, 0, 0, 1, "1: strip / chain kernels whatever the number of tasks"),
    SDF_CF("SDF_STRIP_COLS", strip_cols, 0, 0, 8, "columns per lane of the chained strips (4, 8); 0: by the chunk"),
    SDF_CF("SDF_CHAIN_MIN", chain_min, 3072, 0, 1e12, "chain wavefronts a chunk must hold for the chained strips"),
    SDF_CF("SDF_NO_LANE", no_lane, 0, 0, 1, "1: no lane kernel (small full-band tasks stay on the window kernels)"),
    SDF_CF("SDF_LANE_MIN", lane_min, 8192, 1, 1e12, "eligible tasks a batch must hold for the lane kernel"),
    SDF_CF("SDF_LANE_PLAN", lane_plan_sort, 0, 0, 1, "1 (or 'sort'): lane planning by hipCUB radix sort + scans (round 4) instead of the counting sort"),
    SDF_CF("SDF_LANE_PRIO", lane_prio, 1, 0, 1, "0: the lane stream on the default queue priority"),
    // ---- how a batch is cut and planned ----
    SDF_CF("SDF_PIPELINE", pipeline, 1, 0, 1, "0: one chunk on one stream (isolated kernel timing)"),
    SDF_CF("SDF_CUT_NCH", cut_chunks, 0, 0, 64, "chunks of a pipelined batch; 0: by its size"),
    SDF_CF("SDF_HEAVY_BYTES", heavy_bytes, 0, 0, 1e15, "direction-flag bytes from which a task is heavy; 0: 256 KB"),
    SDF_CF("SDF_EARLY_HEAVY", early_heavy, 1, 0, 1, "0: heavy chunks wait for the whole cut"),
    SDF_CF("SDF_SPLIT_MIN", split_min, -1, -1, 1e12, "tasks from which a batch starts in two parts; -1: the default rule, 0: never"),
    SDF_CF("SDF_SPLIT_DIV", split_div, 8, 2, 1024, "the first part is 1/N of the batch"),
    SDF_CF("SDF_PLAN_THREADS", plan_threads, -1, -1, 15, "planning threads; -1: by the CPUs the process may use"),
    SDF_CF("SDF_PLAN_POOL_FROM", plan_pool_from, 120000, 0, 1e12, "tasks from which parked threads plan the chunks"),
    SDF_CF("SDF_SCAN_POOL_FROM", scan_pool_from, 120000, 0, 1e12, "tasks from which they scan the cut as well"),
    SDF_CF("SDF_POOL_SPIN_US", pool_spin_us, -1, -1, 1e6, "microseconds a parked planning thread spins before it sleeps; -1: default"),
    SDF_CF("SDF_PIN_REGISTER", pin_register, 1, 0, 2, "pinned staging sized by sdf_reserve / sdf_pool_host: 0 hipHostMalloc; 1 the small staging buffers as registered huge pages of the process's own, the character pool and the anchors' staging by hipHostMalloc; 2 all of them registered (fast to get; every copy pays for mapping the pages when they are not huge ones)"),
    {"SDF_WORKSPACE_GIB", "workspace_gib", offsetof(sdf_config, workspace_gib), true, 0, 0, 1e6, "direction-flag workspace, overrides the caller's figure; 0: the caller's"},
    // ---- other entry points ----
    SDF_CF("SDF_CHAIN_THREADS", chain_threads_only, 0, 0, 1, "sdf_chain_batch: 1: every pair on the thread-per-pair kernel"),
    SDF_CF("SDF_STATS_ITEMS", stats_items, 1 << 18, 1, 1e9, "sdf_stats_columns: capacity of the long alignments' segment list"),
    SDF_CF("SDF_STATS_GROUP_MAX", stats_group_max, -1, -1, 1e6, "sdf_stats_columns: runs above which an alignment gets a wavefront of its own; -1: default"),
    // ---- what the library says on stderr ----
    SDF_CF("SDF_DEBUG_PLAN", debug_plan, 0, 0, 1, "1: the cut, the plan and this configuration"),
    SDF_CF("SDF_DEBUG_TIMING", debug_timing, 0, 0, 1, "1: milliseconds of the phases of the entry points"),
    SDF_CF("SDF_DEBUG_CLASSES", debug_classes, 0, 0, 1, "1: tasks, rows and cells of every launch class"),
    SDF_CF("SDF_DEBUG_PLAN_EARLY", debug_plan_early, 0, 0, 1, "sdf_debug_plan: the cut in two passes with the early start"),
};
#undef SDF_CF
constexpr size_t kConfigCount = sizeof(kConfigFields) / sizeof(kConfigFields[0]);

static void config_store(sdf_config *c, const ConfigField &f, double v) {
  if (f.real) *reinterpret_cast<double *>(reinterpret_cast<char *>(c) + f.offset) = v;
  else *reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c) + f.offset) = (int64_t)v;
}
static double config_load(const sdf_config *c, const ConfigField &f) {
  if (f.real) return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(c) + f.offset);
  return (double)*reinterpret_cast<const int64_t *>(reinterpret_cast<const char *>(c) + f.offset);
}

}  // namespace sdf

extern "C" void sdf_config_default(sdf_config *c) {
  if (!c) return;
  memset(c, 0, sizeof *c);
  c->size = (uint32_t)sizeof(sdf_config);
  for (size_t k = 0; k < sdf::kConfigCount; ++k) sdf::config_store(c, sdf::kConfigFields[k], sdf::kConfigFields[k].def);
}

// name: an SDF_* variable or the field's own name; value: a number (SDF_LANE_PLAN also takes "sort" / "bins").
// Unknown name, not a number, out of range: SDF_ERR_INVALID and a message.
extern "C" int sdf_config_set(sdf_config *c, const char *name, const char *value, char *err, size_t errcap) {
  auto fail = [&](const std::string &m) {
    if (err && errcap) snprintf(err, errcap, "%s", m.c_str());
    return SDF_ERR_INVALID;
  };
  if (!c || !name || !value) return fail("sdf_config_set: null argument");
  if (c->size != sizeof(sdf_config)) return fail("sdf_config_set: the struct was not initialised by sdf_config_default (size field)");
  for (size_t k = 0; k < sdf::kConfigCount; ++k) {
    const sdf::ConfigField &f = sdf::kConfigFields[k];
    if (strcmp(name, f.env) != 0 && strcmp(name, f.name) != 0) continue;
    double v;
    if (!strcmp(f.name, "lane_plan_sort") && (value[0] == 's' || value[0] == 'b')) {
      v = value[0] == 's' ? 1 : 0;
    } else {
      char *end = nullptr;
      v = strtod(value, &end);
      if (end == value || (end && *end != 0)) return fail(std::string(f.env) + "=" + value + ": not a number");
    }
    if (v < f.lo || v > f.hi) {
      char range[96];
      snprintf(range, sizeof range, " (allowed: %.15g .. %.15g)", f.lo, f.hi);
      return fail(std::string(f.env) + "=" + value + ": out of range" + range);
    }
    sdf::config_store(c, f, v);
    return SDF_OK;
  }
  return fail(std::string("unknown setting ") + name);
}

extern "C" int sdf_config_from_env(sdf_config *c, char *err, size_t errcap) {
  sdf_config_default(c);
  for (size_t k = 0; k < sdf::kConfigCount; ++k)
    if (const char *e = getenv(sdf::kConfigFields[k].env)) {
      // (flags that used to be "set at all": SDF_DEBUG_* with any value mean 1; any other variable set to nothing is not set)
      const bool flag = !strncmp(sdf::kConfigFields[k].env, "SDF_DEBUG_", 10);
      if (!*e && !flag) continue;
      char *end = nullptr;
      (void)strtod(e, &end);
      const bool numeric = end != e && end && *end == 0;
      if (int rc = sdf_config_set(c, sdf::kConfigFields[k].env, flag && !numeric ? "1" : e, err, errcap)) return rc;
    }
  return SDF_OK;
}

// one "NAME=value" per line, every field; returns the bytes needed (snprintf convention)
extern "C" size_t sdf_config_dump(const sdf_config *c, char *buf, size_t cap) {
  std::string s;
  char line[256];
  for (size_t k = 0; c && k < sdf::kConfigCount; ++k) {
    const sdf::ConfigField &f = sdf::kConfigFields[k];
    const double v = sdf::config_load(c, f);
    snprintf(line, sizeof line, "%s=%.15g%s\n", f.env, v, v != f.def ? "   # (not the default)" : "");
    s += line;
  }
  if (buf && cap) snprintf(buf, cap, "%s", s.c_str());
  return s.size() + 1;
}

extern "C" size_t sdf_config_describe(char *buf, size_t cap) {
  std::string s;
  char line[512];
  for (size_t k = 0; k < sdf::kConfigCount; ++k) {
    const sdf::ConfigField &f = sdf::kConfigFields[k];
    snprintf(line, sizeof line, "%-22s %-20s default %-10.15g %s\n", f.env, f.name, f.def, f.doc);
    s += line;
  }
  if (buf && cap) snprintf(buf, cap, "%s", s.c_str());
  return s.size() + 1;
}
