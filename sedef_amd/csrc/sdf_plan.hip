// Batch planning for the DP entry points (host code only; no device work happens in this file).
//
//   cut_batch()   one pass over the caller's tasks: validation, upper bounds of the direction-flag bytes, the split of
//                 "heavy" tasks from the chunk rotation, chunk boundaries, and -- so that chunks can be planned
//                 independently of each other, on several threads -- every chunk's bases into the shared arrays
//                 (PlanTask index, launch-order index, CIGAR staging word);
//   plan_chunk()  one chunk: kernel choice per task, pairing of tasks of equal geometry, the direction-flag layout
//                 inside the chunk's workspace region, launch classes and the launch order.  Pure function of the
//                 tasks and the PlanEnv: the unit tests of the planner need no GPU.
//
// Replaces the per-call set-up of ksw_extz2_sse (reference: extern/ksw2_extz2_sse.cc:57-98: early-outs, band and
// matrix geometry, allocation) for a batch of tasks.
#include <unordered_map>

#include "sdf_ctx.h"

#include <atomic>
#include <memory>
#include <thread>

namespace sdf {

inline const sdf_config *default_config() {  // (a PlanEnv nobody gave a context's settings: the library's defaults)
  static const sdf_config c = [] {
    sdf_config d;
    sdf_config_default(&d);
    return d;
  }();
  return &c;
}

struct PlanEnv {
  const sdf_config *cfg = default_config();  // the context's settings (sdf_config.hip); the fields below are derived per call
  const sdf_task *tasks = nullptr;
  size_t n = 0;
  uint32_t want = 0;
  bool want_cigar = false;
  bool degenerate = false;  // scoring for which the reference returns before any work (:81)
  int gapo = 0;
  int max_dyn_lds = 64 * 1024;
  bool force_general = false, no_pair = false, no_stripe = false;
  size_t self_pair_max = 512;  // more tasks without a partner than this in a chunk: the wave kernel instead of pairing them with themselves (SDF_SELF_PAIR_MAX)
  // mixed pairs (extz2_pair.hip, MIXED): banded tasks of one (w, flag) and different lengths two to a wavefront -- a
  // throughput kernel: a chunk must hold `mixed_min` candidates (SDF_MIXED_MIN), and then its long banded tasks whose
  // band reaches the end take it too, instead of the banded stripe kernel (SDF_NO_MIXED=1: off)
  bool no_mixed = false;
  size_t mixed_min = 4096;
  int stripe_min = 400;
  int bstripe_min_rows = 4000;  // banded tasks with at least this many anti-diagonals: banded stripe kernel (0: off)
  // lane kernel (extz2_lane.hip): small full-band tasks leave the host's planning altogether when the batch holds at least
  // `lane_min` of them (fewer do not fill the device: a lane walks its matrix alone, ~100 cycles per cell)
  bool strip_ok = false;    // strip kernel (extz2_strip.hip): the scoring is tame
  bool strip_always = false;  // SDF_STRIP_ALWAYS=1 (tests): ... whatever the number of tasks
  int strip_cols = 0;         // SDF_STRIP_COLS (tests): columns per lane of the chained strips
  size_t chain_min = 3072;    // chain wavefronts a chunk must hold for the chained strips (SDF_CHAIN_MIN)
  bool lane_ok = false;     // the scoring is tame and only CIGAR / score / mte are wanted
  size_t lane_min = 8192;
  LaneRec *lane_recs = nullptr;  // pinned, one per task of the batch: filled by the scan
};

struct Launch {
  int bs;  // 64 / 256 / 1024: general kernel with that many threads (+2000: PLAIN flavour); 1, 2, 3, 4, 6, 8: wave kernel with
           // NREG (+10: streamed windows); 100 + NREG: pair kernel (+10: streamed; 120 + NREG: TRACK flavour);
           // 300 + NREG: stripe kernel, 400 + NREG: banded stripe kernel (one launch-order entry per stripe);
           // 1000 / 1001 / 2001: general kernel with its state in HBM
  size_t lds;  // dynamic LDS bytes of the launch (HBM-state classes: slab bytes per workgroup)
  size_t off, cnt;  // entries of the chunk's launch order
  double est;       // duration estimate: the launch's longest task
  int rmax = 0;     // stripe kernel: anti-diagonals (qlen + tlen) of the launch's longest task
};

struct ChunkPlan {
  size_t s = 0, e = 0;   // ordinary chunk: task range [s, e) of the caller's array, minus the heavy tasks in it;
  bool heavy = false;    // heavy chunk: range [s, e) of BatchCut::heavy_idx
  size_t pb = 0;         // first PlanTask of the chunk
  size_t ob = 0;         // first launch-order entry (room for `order_cap`: a task paired with itself is listed twice,
                         // a stripe task once per stripe)
  size_t order_cap = 12288;  // (up to five stripe launches -- one width of the full-band kernel, three of the banded one -- of up
                             // to 8 x 254 idle entries each)
  int64_t stage0 = 0;    // first CIGAR staging word
  size_t ntask = 0;      // tasks the chunk will plan (known after cut_batch)
  int64_t stage_words = 0;  // CIGAR staging words of those tasks
  // filled by plan_chunk
  size_t cnt = 0, nord = 0;
  std::vector<Launch> launches;
  unsigned layouts = 0;  // direction-flag layouts present: bit 0 byte rows, 1 wave blocks, 2 pair blocks, 3 stripes
  long long paired = 0;
  size_t dir_bytes = 0;
  const char *err = nullptr;
};

#define SDF_CUT_BLOCK 4096

struct BatchCut {
  std::vector<ChunkPlan> chunks;  // heavy chunks first
  std::vector<uint8_t> heavy;     // per task, when split_heavy
  std::vector<uint32_t> heavy_idx;  // the heavy tasks, ascending
  // scratch of cut_batch, kept by the context between calls (fresh vectors of this size cost a millisecond of page faults)
  std::vector<uint32_t> bound;      // per task: upper bound of its direction flags, in units of 256 bytes
  std::vector<uint32_t> cap;        // per task: CIGAR staging words | 0x80000000 when the task runs at all
  std::vector<uint32_t> hparts[16];  // heavy task indices, per scan thread
  std::vector<sdf_task> htasks[16];  // ... and their records: plan_chunk reads the heavy tasks from a compact copy (they lie
                                     // scattered over the caller's array -- a cache miss each, on the thread in front of the
                                     // call's first launch)
  std::vector<sdf_task> heavy_tasks;  // the records of heavy_idx, in its order
  struct Block {  // sums over SDF_CUT_BLOCK consecutive tasks: all runnable ones / the heavy ones / the lane tasks among them
    uint64_t bd = 0, hbd = 0;  // direction-flag bounds, bytes
    uint32_t nt = 0, hnt = 0, sw = 0, hsw = 0, oc = 0, hoc = 0;  // tasks, CIGAR staging words, launch-order entries
    uint32_t lnt = 0, lsw = 0;  // lane tasks (not in the sums above) and their staging words
    uint64_t ldir = 0;          // direction-flag bytes of the lane tasks in the lane kernel's own layout
    uint32_t lcls[4] = {0, 0, 0, 0};  // lane tasks per launch class (query length)
  };
  std::vector<Block> blocks;
  void reset() {
    chunks.clear();
    heavy_idx.clear();
    split_heavy = pipelined = false;
    nch = max_regions = nreg_ws = 1;
    n_heavy = 0;
    region_need = 16;
    heavy_need = 0;
    stage_total = 0;
    ntask_total = 0;
    order_total = 0;
    use_lane = false;
    n_early = 0;
    stage_upper = 0;
    order_upper = 0;
    n_lane = 0;
    lane_stage_words = 0;
    lane_dir_bytes = 0;
    for (auto &c : lane_cls) c = 0;
  }
  // lane kernel: tasks the scan found eligible (lane[k] != 0), taken out of the chunks when there are enough of them
  std::vector<uint8_t> lane;
  bool use_lane = false;
  size_t n_lane = 0, lane_cls[4] = {0, 0, 0, 0};
  int64_t lane_stage_words = 0;
  size_t lane_dir_bytes = 0;
  bool split_heavy = false, pipelined = false;
  size_t nch = 1, max_regions = 1, n_heavy = 0;
  size_t region_need = 16, heavy_need = 0, nreg_ws = 1;
  int64_t stage_total = 0;
  size_t ntask_total = 0;
  size_t order_total = 0;
  // early start (cut_batch's `early` callback): the heavy chunks are cut after a first pass over the BIG tasks only and
  // handed to the caller -- which plans and launches them -- while the pass over the rest of the batch runs
  size_t n_early = 0;         // chunks of `chunks` (its first ones, all heavy) that the callback has launched
  int64_t stage_upper = 0;    // upper bound of stage_total + the lane tasks' words, known after the first pass
  size_t order_upper = 0;     // upper bound of order_total
};

namespace plan_detail {

// stripe width of the banded stripe kernel: the narrowest with at most 254 stripes (0: target too long)
inline int bstripe_nreg(int tlen, const int forced) {  // forced: sdf_config.bstripe_nreg (tests: wider stripes than the target needs)
  const int t16 = (tlen + 15) / 16 * 16;
  if ((forced == 2 || forced == 4) && t16 <= 254 * 128 * forced) return forced;
  return t16 <= 254 * 128 ? 1 : t16 <= 254 * 256 ? 2 : t16 <= 254 * 512 ? 4 : 0;
}

inline bool task_runs(const sdf_task &t, bool degenerate) { return t.qlen > 0 && t.tlen > 0 && !degenerate; }

struct Cls {
  int bs;
  size_t lds;       // class key
  size_t need_max;  // largest real requirement in the class: what the launch asks for
  std::vector<int32_t> idx;
  double est = 0;
};

}  // namespace plan_detail

// Per-thread scratch of plan_chunk (kept between chunks: no allocation in the steady state).
struct PlanScratch {
  std::vector<int32_t> win_need, partner;
  std::vector<std::pair<int32_t, int32_t>> table;
  std::vector<plan_detail::Cls> cls;
  std::vector<char> tracked, mixedf;
  std::vector<int32_t> bs_alt;       // tasks given to the banded stripe kernel that a mixed pair could take: index, wave nreg, window need
  std::vector<uint64_t> mix_keys;
  std::vector<int32_t> stripe_lane, stripe_fill;
  std::vector<uint64_t> strip_keys, strip_keys_tmp;
};

// Returns SDF_OK or an error code with *err set.
// `early` (optional): called once the heavy chunks are known -- cut.chunks holds them, with their bases; cut.heavy_need,
// stage_upper, order_upper are set -- while the rest of the batch is still being read: it plans and launches them and
// sets cut.n_early (batches of 400,000 tasks and more on a context with planning threads: the long tasks of a batch are
// the launch that ends it, and the pass over a million task records is 1-2 ms it need not wait for).
static int cut_batch(const PlanEnv &env, bool pipeline_enabled, size_t ws_budget, BatchCut &cut, const char **err,
                     WorkerPool *pool = nullptr, const std::function<int()> *early = nullptr) {
  const sdf_task *tasks = env.tasks;
  const size_t n = env.n;
  // The batch is cut into chunks that are planned, uploaded and launched one after the other: while the GPU runs
  // chunk i the host plans the following ones, the big DP launches of consecutive chunks alternate between two streams
  // (the next chunk fills the CUs while the previous one drains), launches of a few tasks (each a full task latency
  // long) go, longest first, to whichever stream has the least work queued, and the traceback of a chunk runs next to
  // the following chunk's DP.  Direction-flag regions rotate over `nreg_ws` slices of the workspace.
  // (small batches stay on the caller's stream -- unless they hold long tasks: their launch classes, each as long
  // as its longest task, then run side by side on the other streams like those of a large batch)
  bool any_long = false;
  if (n < 2048)
    for (size_t k = 0; k < n && !any_long; ++k) any_long = tasks[k].qlen + (int64_t)tasks[k].tlen >= 3000;
  cut.pipelined = pipeline_enabled && (n >= 2048 || any_long);
  cut.nch = 1;
  // (sixteen or twenty-four chunks for a million tasks were measured no better than eight)
  // (two, three or six chunks for the 100,000-task headline batch: within noise of four)
  if (cut.pipelined && n >= 32768) cut.nch = std::min<size_t>(n >= 500000 ? 8 : 4, n / 16384);
  const int force_nch = (int)env.cfg->cut_chunks;  // (experiments: the number of ordinary chunks a large batch is cut into)
  if (force_nch > 0 && cut.pipelined) cut.nch = (size_t)force_nch;
  cut.max_regions = cut.nch > 4 ? 8 : cut.nch > 1 ? 4 : 1;
  bool chain_bound = false;  // (set after the scan: fewer chunks than the size of the batch alone would give)
  const size_t nch_by_size = cut.nch;
  // the first chunk is small, so that the GPU starts early, but fills the wavefront slots of the device (4,096 pairs
  // of tasks of the headline shape) while the next chunks are being planned
  const size_t first_target0 = nch_by_size > 1 ? std::max<size_t>(8192, n / (16 * nch_by_size + 1)) : n;

  // A task is "heavy" when its wavefront (or workgroup) is busy for about a millisecond or more whatever else runs:
  // 500 x 500 and up at full band, i.e. from 256 KB of direction flags.
  const size_t heavy_min = env.cfg->heavy_bytes > 0 ? (size_t)env.cfg->heavy_bytes : (size_t)256 << 10;
  // ---- validation + upper bound of each task's direction flags, whichever kernel takes it ----
  // (on several threads for batches of hundreds of thousands of tasks: this pass and the loop below are all the
  // planning the GPU waits for besides the first chunk)
  const bool dbg_cut = env.cfg->debug_plan != 0;
  const auto tc0 = std::chrono::steady_clock::now();
  std::vector<uint32_t> &bound = cut.bound, &cap = cut.cap;
  const bool lane_scan = env.lane_ok && env.lane_recs && n >= env.lane_min;
  if (lane_scan && cut.lane.size() < n) cut.lane.resize(n);
  // (not cleared: 8 MB of memset per million tasks, on this thread, before anything else can start -- the scan below
  // writes both words of every task, runnable or not)
  if (bound.size() < n) bound.resize(n);
  if (cap.size() < n) cap.resize(n);
  size_t heavy_bytes = 0, heavy_budget = 0;
  std::vector<ChunkPlan> heavy_chunks;
  std::vector<uint32_t> (&hparts)[16] = cut.hparts;
  for (auto &hp : hparts) hp.clear();
  for (auto &ht : cut.htasks) ht.clear();
  cut.heavy_tasks.clear();
  const size_t nblk = (n + SDF_CUT_BLOCK - 1) / SDF_CUT_BLOCK;
  cut.blocks.assign(nblk, BatchCut::Block());
  auto banded_long = [&](const sdf_task &t) {  // (a superset of what plan_chunk gives to the banded stripe kernel)
    return env.bstripe_min_rows > 0 && t.w >= 1 && t.qlen + t.tlen - 1 >= env.bstripe_min_rows && plan_detail::bstripe_nreg(t.tlen, (int)env.cfg->bstripe_nreg) > 0;
  };
  auto order_entries = [&](const sdf_task &t) -> uint32_t {  // a task paired with itself is listed twice, a stripe task
    // (chained strips take full-band targets of 513..65536 bases whatever stripe_min is: eight idle entries of padding per
    // block at most -- ADVICE r3: with SDF_STRIPE_MIN above 512 their launch order overflowed)
    const bool chain = env.strip_ok && t.tlen > kStripMaxT && t.tlen <= kStripChainMaxT;
    return 2 + (t.tlen > env.stripe_min && t.tlen <= kStripeMaxT ? (uint32_t)(t.tlen + 127) / 128 : chain ? 8u * (uint32_t)strip_blocks(t.tlen, 4) : 0u) +  // once per stripe
           (banded_long(t) ? (uint32_t)(t.tlen + 15 + 127) / 128 : 0u);
  };
  {
    // (what a task needs depends on (qlen, tlen, w, SCORE_ONLY) alone, and batches repeat their geometries -- the headline
    // batch has a hundred of them in 100,000 tasks: a small direct-mapped memo per scan thread)
    struct Memo {
      int32_t qlen = -1, tlen = -1, w = 0, so = 0;
      uint32_t oc = 0, words = 0;
      size_t bd = 0, hvb = 0;
    };
    struct Part {
      size_t nh = 0, hb = 0;
      bool bad = false;
      Memo memo[256];
      int64_t words = 0;       // first pass: q + t + 2 of every task, the big tasks' launch-order entries, their number
      size_t big_oc = 0, nbig = 0;
      int64_t rows_max = 0;    // anti-diagonals of the longest task accounted for
    };
    const size_t hv_limit = n / 4 + 1;
    // what a task needs of the chunks' budgets -- flag bytes whichever window kernel takes it, staging words, launch-order
    // entries -- into its block's sums
    auto account = [&](size_t k, Part &pt, std::vector<uint32_t> &hv, const bool may_be_heavy = true) {
      const sdf_task &t = tasks[k];
      BatchCut::Block &blk = cut.blocks[k / SDF_CUT_BLOCK];
      cap[k] = 0x80000000u;
      pt.rows_max = std::max(pt.rows_max, (int64_t)t.qlen + t.tlen);
      Memo &mm = pt.memo[((uint32_t)t.qlen * 31u + (uint32_t)t.tlen * 17u + (uint32_t)t.w) & 255u];
      const int32_t so = t.flag & SDF_FLAG_SCORE_ONLY;
      size_t bd = 0, hvb = 0;
      uint32_t oc, words;
      if (mm.qlen == t.qlen && mm.tlen == t.tlen && mm.w == t.w && mm.so == so) {
        oc = mm.oc, words = mm.words, bd = mm.bd, hvb = mm.hvb;
        ++blk.nt;
        blk.oc += oc;
        cap[k] |= words;
        blk.sw += words;
      } else {
      oc = order_entries(t);
      ++blk.nt;
      blk.oc += oc;
      // A task that wants no CIGAR needs no direction flags -- except on the stripe kernels, whose progress words,
      // hand-over values and edge columns live in HBM right behind the task's flag blocks: those tasks reserve the
      // whole layout whatever they want.
      const bool with_dir = env.want_cigar && !(t.flag & SDF_FLAG_SCORE_ONLY);
      words = with_dir ? (uint32_t)(t.qlen + t.tlen + 2) : 0u;
      cap[k] |= words;
      blk.sw += words;
      const int w = t.w < 0 ? std::max(t.qlen, t.tlen) : t.w;
      const int ncol16 = ((std::min(std::min(t.qlen, t.tlen), w + 1) + 15) / 16 + 1) * 16;
      const size_t nrow = (size_t)t.qlen + t.tlen - 1;
      const int need = std::min(ncol16 + 32, (t.tlen + 15) / 16 * 16);
      if (with_dir) {
        // (the reference's byte per cell is the general kernel's layout: a task with nothing special asked of it, whose band
        // reaches the end and whose window a register-resident kernel holds never goes there -- plan_chunk's `plain_ok`;
        // for a batch of long banded tasks the byte rows are twice the bits, and the bound is what cuts it into chunks)
        bool bits_only = false;
        if (need <= 1024 && !env.force_general && !(env.want & SDF_WANT_EXT) && t.zdrop < 0 && env.gapo >= 0 &&
            !(t.flag & (SDF_FLAG_RIGHT | SDF_FLAG_EXTZ_ONLY | SDF_FLAG_GENERIC_SC | SDF_FLAG_APPROX_MAX | SDF_FLAG_APPROX_DROP)) &&
            (w >= 1 || nrow == 1)) {
          Band bl;
          const int wn = need <= 128 ? 1 : need <= 256 ? 2 : need <= 384 ? 3 : need <= 512 ? 4 : need <= 768 ? 6 : 8;
          bits_only = band_of((int)nrow - 1, t.qlen, t.tlen, w, bl) && wave_lds_bytes(t.qlen, t.tlen, wn) <= (size_t)env.max_dyn_lds;
        }
        hvb = (nrow * (size_t)ncol16 + 16 + 255) & ~(size_t)255;  // (what "heavy" is measured by, whichever layout the task takes)
        // (round 6) ... nor does a wide FULL-BAND task that plan_chunk always hands to the stripe / strip / chain kernels (their
        // own bounds follow below: 0.5-0.8 bytes per cell against the byte rows' two per cell of a square matrix).  The
        // far-gap round of the chr1-sized stage -- 10,813 tasks, 9.5e9 cells -- was cut into two chunks of one 7 ms chain each
        // by that bound; one chunk: DP 14.4 -> see profiles/r06_stage_dp2.txt.
        const bool striped_always = !env.force_general && !env.no_stripe && !(env.want & SDF_WANT_EXT) && t.zdrop < 0 && env.gapo >= 0 &&
                                    !(t.flag & (SDF_FLAG_RIGHT | SDF_FLAG_EXTZ_ONLY | SDF_FLAG_GENERIC_SC | SDF_FLAG_APPROX_MAX | SDF_FLAG_APPROX_DROP)) &&
                                    w >= std::max(t.qlen, t.tlen) && t.tlen > env.stripe_min && t.tlen <= kStripeMaxT &&
                                    stripe_lds_bytes(t.qlen, 4) <= (size_t)env.max_dyn_lds;
        if (!bits_only && !striped_always) bd = hvb;
        // (the wave kernel's blocks: 1 KiB per register of 128 slots -- 1, 2, 3, 4, 6 or 8 of them; the pair kernels' 512 B per
        // register of 64 slots, up to nine, stay below)
        if (need <= 1024)
          bd = std::max(bd, (nrow + 15) / 16 * (size_t)(need <= 512 ? (need + 127) / 128 : need <= 768 ? 6 : 8) * 1024);
        // (the pair kernel's TRACK flavour rounds its window to 3 / 6 registers of 64 slots, 512 B each per block)
        if (need <= 384) bd = std::max(bd, (nrow + 15) / 16 * (size_t)(need <= 192 ? 3 : 6) * 512);
        // (a MIXED pair gives both tasks the register count of the wider window: a banded task of a short target may sit next
        // to one whose window is as wide as the band allows -- ((w + 1 + 15) / 16 + 1) * 16 + 32 slots, at most kMixedMaxNeed --
        // at 512 B per register and block of its OWN rows)
        if (bits_only && need <= kMixedMaxNeed && w < std::max(t.qlen, t.tlen)) {
          const int wide = std::min(((w + 1 + 15) / 16 + 1) * 16 + 32, kMixedMaxNeed);
          const int regs = (wide + 63) / 64;
          const int nreg = regs <= 2 ? 2 : regs <= 6 ? regs : regs <= 8 ? 8 : 9;
          bd = std::max(bd, (nrow + 15) / 16 * (size_t)nreg * 512);
        }
      }
      if (t.tlen > env.stripe_min && t.tlen <= kStripeMaxT && w >= std::max(t.qlen, t.tlen))  // (stripe kernel, any width)
        for (int nr = 1; nr <= 4; nr *= 2)
          bd = std::max(bd, (stripe_dir_bytes(t.qlen, t.tlen, nr) + stripe_sync_bytes(t.qlen, t.tlen, nr) + 255) & ~(size_t)255);
      if (banded_long(t)) {
        const int nr = plan_detail::bstripe_nreg(t.tlen, (int)env.cfg->bstripe_nreg);
        bd = std::max(bd, (bstripe_dir_bytes(t.qlen, t.tlen, w, nr) + bstripe_sync_bytes(t.qlen, t.tlen, w, nr) + 255) & ~(size_t)255);
      }
      // (strip kernel: a region per PAIR of tasks with as many column blocks, sized by the one with more rows -- which
      // this pass does not know.  A pair of m and M <= 1.125 m + 64 rows needs blocks x (M + 63) records of 8 bytes, a task
      // without a partner blocks x (rows + 63) records of 4; every task reserves blocks x (0.54 rows + 60) x 520 bytes:
      // enough for either, the chains' edge columns included -- 0.54 (1.889 M - 57) + 120 >= M + 64.  Until the end of
      // round 6 the pairs were allowed M <= 1.5 m + 64 and the factor was 0.8: the far-gap round of the chr1-sized stage
      // reserved 9.1 GB for flags that take 5.0, and a workspace of 8 GiB cut it in two)
      // (wider targets: a chain of wavefronts, the edge columns between the blocks behind the records -- reserved, like
      // the other stripe kernels' inter-stripe words, whatever the task wants)
      if (env.strip_ok && t.tlen > 256 && t.tlen <= kStripChainMaxT && w >= std::max(t.qlen, t.tlen) &&
          (with_dir || t.tlen > kStripMaxT))
        // (a chain's blocks are four columns per lane wide when its chunk holds few of them -- plan_chunk decides, unless
        // SDF_STRIP_COLS=8 does --: twice the blocks, a record per PAIR of steps -- the same bytes, one block of rounding
        // apart, whichever plan_chunk picks)
        bd = std::max(bd, t.tlen > kStripMaxT && env.strip_cols != 8
                              ? ((size_t)std::max(strip_blocks(t.tlen, 4), 2 * strip_blocks(t.tlen, 8)) *
                                     (size_t)(t.qlen * 27 / 50 + 60) * 260 + 512 + 255) & ~(size_t)255
                              : ((size_t)strip_blocks(t.tlen, 8) * (size_t)(t.qlen * 27 / 50 + 60) * 520 + 512 + 255) & ~(size_t)255);
      mm.qlen = t.qlen, mm.tlen = t.tlen, mm.w = t.w, mm.so = so;
      mm.oc = oc, mm.words = words, mm.bd = bd, mm.hvb = hvb;
      }
      bound[k] = (uint32_t)std::min<size_t>(bd >> 8, 0xffffffffu);
      blk.bd += (uint64_t)bound[k] << 8;
      if (std::max(bd, hvb) >= heavy_min && may_be_heavy) {
        ++pt.nh;
        pt.hb += bd;
        ++blk.hnt;
        blk.hbd += (uint64_t)bound[k] << 8;
        blk.hsw += words;
        blk.hoc += oc;
        if (pt.nh <= hv_limit) {  // (beyond a quarter of the batch there is no split)
          hv.push_back((uint32_t)k);
          cut.htasks[&hv - &hparts[0]].push_back(t);
        }
      }
    };
    // mode 0: every task; 1: the big tasks only (and the sums of the early start); 2: the others, none of them heavy
    // (big: everything a stripe / strip kernel may take -- more than 256 target bases --, and from 40,000 cells or 1,000
    // anti-diagonals; the others have two launch-order entries and no direction-flag bound near heavy_min)
    auto is_big = [&](const sdf_task &t) {
      return t.tlen > 256 || (int64_t)t.qlen * t.tlen >= 40000 || (int64_t)t.qlen + t.tlen >= 1000 || banded_long(t);
    };
    auto scan = [&](size_t lo, size_t hi, Part &pt, std::vector<uint32_t> &hv, const int mode) {  // (lo: a multiple of the block size)
      for (size_t k = lo; k < hi; ++k) {
        const sdf_task &t = tasks[k];
        if (mode == 1) {
          pt.words += (int64_t)t.qlen + t.tlen + 2;
          if (!is_big(t)) continue;
          ++pt.nbig;
          pt.big_oc += order_entries(t);
        } else if (mode == 2 && is_big(t)) {
          continue;
        }
        if (t.flag & 0x300) {  // (not KSW_EZ_* bits of the extz2 kernel)
          pt.bad = true;
          return;
        }
        bound[k] = 0;
        cap[k] = 0;
        if (lane_scan) {
          cut.lane[k] = 0;
          env.lane_recs[k].flag = 0xffffu;
        }
        if (!plan_detail::task_runs(t, env.degenerate)) continue;
        // a small full-band task with nothing special asked of it is the lane kernel's (if the batch turns out to hold
        // enough of them: else these tasks are accounted for in a second pass, below): sixteen bytes for the device
        // and four sums, nothing else
        if (lane_scan && t.qlen <= kLaneMaxLen && t.tlen <= kLaneMaxLen && (int64_t)t.qlen * t.tlen <= kLaneMaxCells &&
            (t.w < 0 || t.w >= std::max(t.qlen, t.tlen)) && t.zdrop < 0 && !(t.flag & ~(SDF_FLAG_SCORE_ONLY | SDF_FLAG_REV_CIGAR)) &&
            (uint64_t)t.q_off < 0xffffffffull && (uint64_t)t.t_off < 0xffffffffull) {
          const bool with_dir = env.want_cigar && !(t.flag & SDF_FLAG_SCORE_ONLY);
          BatchCut::Block &blk = cut.blocks[k / SDF_CUT_BLOCK];
          cut.lane[k] = 1;
          LaneRec &lr = env.lane_recs[k];
          lr.q_word = (uint32_t)t.q_off;
          lr.t_word = (uint32_t)t.t_off;
          lr.out_idx = (uint32_t)k;
          lr.qlen_m1 = (uint8_t)(t.qlen - 1);
          lr.tlen_m1 = (uint8_t)(t.tlen - 1);
          lr.flag = (uint16_t)((t.flag & SDF_FLAG_REV_CIGAR) | (with_dir ? 0 : SDF_FLAG_SCORE_ONLY));
          ++blk.lnt;
          blk.lsw += with_dir ? (uint32_t)(t.qlen + t.tlen + 2) : 0u;
          blk.ldir += with_dir ? lane_dir_bytes(t.qlen, t.tlen) : 0;
          ++blk.lcls[lane_class(t.qlen)];
          continue;
        }
        account(k, pt, hv, mode != 2);
      }
    };
    // (on the context's parked planning threads when there are any; this thread takes a share too)
    // (from 400,000 tasks: on the 100,000-task headline batch four scan threads were measured at 5.2 ms of planning
    // before the first launch against 1.2 ms on this thread alone -- the wake-up of parked threads, not the scan)
    // (the parked threads scan the cut of every batch that has them -- sdf_api.hip: from 120,000 tasks -- : with them the scan
    // runs in two passes and the heavy chunks start after the first, over the big tasks only.  It was 400,000 until round 4:
    // an eighth / a quarter of the hg19 mixture, what a rank of an 8- / 4-GPU strong-scaling run gets, 5.6-5.7 -> 4.9-5.3 ms
    // and 7.2-7.5 -> 5.9-6.0 ms, first launch at 0.56 instead of 1.39 ms.  SDF_SCAN_POOL_FROM overrides.)
    const size_t scan_from = (size_t)env.cfg->scan_pool_from;
    const int nthr = pool && n >= scan_from ? std::min(16, pool->size() + 1) : 1;
    Part parts[16];
    // Runs of sixteen blocks are handed out through a counter: this thread starts at once, a parked helper joins when it
    // has woken up (which takes up to milliseconds on a box with a CPU quota) and takes what is left -- nobody waits
    // for a share that was dealt to a thread still asleep.
    struct Share {
      std::atomic<size_t> next{0}, done{0};
    };
    const size_t run_blocks = 16, nrun = (nblk + run_blocks - 1) / run_blocks;  // (runs of 1 / 2 / 4 blocks: 3.3-4.0 / 2.0-2.7 / 1.8-2.5 ms
                                                                                // per million tasks against 1.3-2.1)
    // one pass over the batch in `mode` (see scan); `meanwhile`: what this thread does first, while the helpers read
    auto run_pass = [&](const int mode, const std::function<void()> *meanwhile) {
      auto share = std::make_shared<Share>();  // (outlives this call: a helper may wake up after everything is done)
      auto work = [&, share, nrun, mode](int q) {  // (touches nothing of this frame once the runs are handed out)
        for (size_t ru = share->next.fetch_add(1); ru < nrun; ru = share->next.fetch_add(1)) {
          if (!parts[q].bad)
            scan(ru * run_blocks * SDF_CUT_BLOCK, std::min(n, (ru + 1) * run_blocks * SDF_CUT_BLOCK), parts[q], hparts[q], mode);
          share->done.fetch_add(1);
        }
      };
      for (int q = 1; q < nthr; ++q) pool->submit([work, q] { work(q); });
      if (meanwhile) (*meanwhile)();
      work(0);
      while (share->done.load() < nrun) std::this_thread::yield();  // (a helper still inside its last run)
    };
    // Heavy tasks leave the chunk rotation when they are a minority: they are planned and launched FIRST, all together
    // (a launch of few long tasks lasts as long as its longest task: one such launch per kernel, not one per chunk), with
    // a workspace slice of their own, and run next to the chunks of ordinary tasks.
    bool two_pass_cut = false;
    auto cut_heavy = [&]() {  // (after the pass that finds them)
      for (int q = 0; q < nthr; ++q) {
        cut.n_heavy += parts[q].nh;
        heavy_bytes += parts[q].hb;
      }
      cut.split_heavy = cut.pipelined && cut.n_heavy * 4 <= n;
      heavy_budget = cut.split_heavy && cut.n_heavy ? std::min(heavy_bytes + 256, ws_budget / 2) : 0;
      // (round 6) a batch whose flags fit the workspace as a whole: the heavy tasks take what they need in ONE chunk -- a
      // launch of few long tasks lasts as long as its longest chain however the rest is cut -- and the others the rest.  Known
      // only where the scan has seen every task (the single pass of a batch below the two-pass threshold).
      if (heavy_budget && !two_pass_cut && heavy_bytes + 256 > heavy_budget) {
        uint64_t all = 0;
        for (const BatchCut::Block &blk : cut.blocks) all += blk.bd;
        const double rest = (double)(all - std::min<uint64_t>(all, heavy_bytes)) * 1.05 + 4096.0;
        if ((double)heavy_bytes + 256 + rest + (double)cut.lane_dir_bytes <= (double)ws_budget) heavy_budget = heavy_bytes + 256;
      }
      cut.heavy.assign(cut.split_heavy ? n : 0, 0);
      if (!cut.split_heavy) return;
      // heavy chunks: ranges of the (ascending) list of heavy tasks
      {  // (the scan threads took their runs of blocks in any order: sorted by task index, the records along)
        std::vector<std::pair<uint32_t, uint32_t>> by_idx;  // (task, position in the concatenation of the threads' lists)
        std::vector<const sdf_task *> rec;
        for (int q = 0; q < 16; ++q)
          for (size_t j = 0; j < hparts[q].size(); ++j) {
            by_idx.push_back({hparts[q][j], (uint32_t)rec.size()});
            rec.push_back(&cut.htasks[q][j]);
          }
        std::sort(by_idx.begin(), by_idx.end());
        cut.heavy_idx.resize(by_idx.size());
        cut.heavy_tasks.resize(by_idx.size());
        for (size_t j = 0; j < by_idx.size(); ++j) {
          cut.heavy_idx[j] = by_idx[j].first;
          cut.heavy_tasks[j] = *rec[by_idx[j].second];
        }
      }
      ChunkPlan hcur;
      hcur.heavy = true;
      size_t hacc = 0;
      for (size_t pos = 0; pos < cut.heavy_idx.size(); ++pos) {
        if (pos + 16 < cut.heavy_idx.size()) {  // (the heavy tasks lie scattered over the batch: a cache miss each)
          const uint32_t kn = cut.heavy_idx[pos + 16];
          __builtin_prefetch(&bound[kn]);
          __builtin_prefetch(&cap[kn]);
        }
        const uint32_t k = cut.heavy_idx[pos];
        const size_t bd = (size_t)bound[k] << 8;
        cut.heavy[k] = 1;
        if (pos > hcur.s && hacc + bd > heavy_budget) {
          hcur.e = pos;
          heavy_chunks.push_back(hcur);
          cut.heavy_need = std::max(cut.heavy_need, hacc);
          hcur = ChunkPlan();
          hcur.heavy = true;
          hcur.s = pos;
          hacc = 0;
        }
        hacc += bd;
        ++hcur.ntask;
        hcur.stage_words += cap[k] & 0x7fffffffu;
        hcur.order_cap += order_entries(cut.heavy_tasks[pos]);
      }
      if (!cut.heavy_idx.empty()) {
        hcur.e = cut.heavy_idx.size();
        heavy_chunks.push_back(hcur);
        cut.heavy_need = std::max(cut.heavy_need, hacc);
      }
      cut.heavy_need = (cut.heavy_need + 255) & ~(size_t)255;
    };
    auto any_bad = [&]() {
      for (int q = 0; q < nthr; ++q)
        if (parts[q].bad) return true;
      return false;
    };
    const bool two_pass = early && *early && nthr > 1 && cut.pipelined;
    two_pass_cut = two_pass;
    int early_rc = SDF_OK;
    if (two_pass) {
      run_pass(1, nullptr);
      if (any_bad()) {
        *err = "unknown task flag";
        return SDF_ERR_UNSUPPORTED;
      }
      size_t nbig = 0, big_oc = 0;
      for (int q = 0; q < nthr; ++q) {
        cut.stage_upper += parts[q].words;
        nbig += parts[q].nbig;
        big_oc += parts[q].big_oc;
      }
      cut.order_upper = big_oc + 2 * (n - nbig) + 64;
      cut_heavy();
      for (int q = 0; q < nthr; ++q) parts[q].nh = parts[q].hb = 0;
      const std::function<void()> start = [&]() {
        if (!cut.split_heavy || heavy_chunks.empty()) return;
        cut.chunks = heavy_chunks;  // (their bases: heavy chunks come first)
        size_t pb = 0, ob = 0;
        int64_t stage = 0;
        for (ChunkPlan &c : cut.chunks) {
          c.pb = pb;
          c.ob = ob;
          c.stage0 = stage;
          pb += c.ntask;
          ob += c.order_cap;
          stage += c.stage_words;
        }
        early_rc = (*early)();
        for (size_t q = 0; q < heavy_chunks.size() && q < cut.chunks.size(); ++q) heavy_chunks[q] = cut.chunks[q];  // (planned)
      };
      run_pass(2, &start);
    } else {
      run_pass(0, nullptr);
    }
    if (early_rc != SDF_OK) {
      *err = "early start of the heavy chunks failed";
      return early_rc;
    }
    if (lane_scan) {
      for (const BatchCut::Block &blk : cut.blocks) {
        cut.n_lane += blk.lnt;
        cut.lane_stage_words += blk.lsw;
        cut.lane_dir_bytes += blk.ldir;
        for (int c = 0; c < 4; ++c) cut.lane_cls[c] += blk.lcls[c];
      }
      // (a quarter of the workspace in 4-bit flags would be more than 10^10 cells of small tasks)
      cut.use_lane = cut.n_lane >= env.lane_min && cut.lane_dir_bytes + 256 <= ws_budget / 4;
      if (!cut.use_lane && cut.n_lane) {  // too few of them after all: they are ordinary tasks
        for (size_t k = 0; k < n; ++k)
          if (cut.lane[k]) {
            cut.lane[k] = 0;
            account(k, parts[0], hparts[0], !two_pass);
          }
        cut.n_lane = 0;
      }
    }
    if (any_bad()) {
      *err = "unknown task flag";
      return SDF_ERR_UNSUPPORTED;
    }
    if (!two_pass) cut_heavy();
    // A launch lasts as long as its longest task -- a chain of qlen + tlen dependent rows, ~0.5 us each when the wavefront
    // shares its SIMD -- whatever else runs: a chunk whose whole work is a few such chains long keeps the device waiting
    // for one wavefront at the end of each of its launches (a batch of banded tasks of all lengths, BASELINE configs[4]:
    // fourteen chunks 400-580 ms, one chunk 220 ms, profiles/r04_shapes.txt).  Such a batch is cut into fewer chunks:
    // each at least six chains' worth of work (the bytes of the direction flags stand for the cells: ~0.65 B each at a
    // Tcell/s), the workspace permitting.
    if (!two_pass && !cut.split_heavy && cut.nch > 1 && !force_nch) {
      int64_t rows_max = 0;
      uint64_t bytes = 0;
      for (int q = 0; q < nthr; ++q) rows_max = std::max(rows_max, parts[q].rows_max);
      for (const BatchCut::Block &blk : cut.blocks) bytes += blk.bd;
      const double chain_ms = (double)rows_max * 0.5e-3, work_ms = (double)bytes / 0.65e9;
      const size_t nch_max = (size_t)std::max(1.0, work_ms / (6.0 * chain_ms));
      if (nch_max < cut.nch) {
        // (no small first chunk either: it would hold a whole region of the workspace for a twelfth of the tasks, and the
        // planning it hides is a few milliseconds of a call of hundreds)
        // Chunks of equal task counts, a workspace region each while the flags of the whole batch fit the workspace; beyond
        // that two regions, the chunks as large as a region holds (two in flight, the third waits for the first's walk).
        const size_t avail = ws_budget - std::min(ws_budget / 2, (size_t)cut.lane_dir_bytes);
        const double need = (double)bytes * 1.05;
        if (need <= (double)avail) {
          cut.nch = nch_max;
          cut.max_regions = std::max<size_t>(1, cut.nch);
        } else {
          cut.max_regions = 2;
          cut.nch = std::max<size_t>(2, (size_t)(need / ((double)avail / 2.0)) + 1);
        }
        chain_bound = true;
      }
    }
  }
  const auto tc1 = std::chrono::steady_clock::now();
  const size_t lane_budget = cut.use_lane ? cut.lane_dir_bytes + 256 : 0;
  const size_t region_budget = (ws_budget - heavy_budget - lane_budget) / cut.max_regions;

  // ---- chunk boundaries ----
  const size_t nch = cut.nch;
  const size_t first_target = chain_bound ? (n + nch - 1) / nch : nch_by_size > 1 ? first_target0 : n;
  const size_t chunk_target = chain_bound ? (n + nch - 1) / nch : nch_by_size > 1 ? (n - first_target + nch - 1) / nch : n;
  std::vector<ChunkPlan> normal;
  {
    // whole blocks of tasks at a time (their sums come from the scan above); task by task only inside a block that
    // does not fit the region as a whole
    ChunkPlan cur;
    size_t acc = 0;
    const uint8_t *hv = cut.split_heavy ? cut.heavy.data() : nullptr;
    const uint8_t *ln = cut.use_lane ? cut.lane.data() : nullptr;
    auto close_at = [&](size_t k) {
      cur.e = k;
      normal.push_back(cur);
      cut.region_need = std::max(cut.region_need, acc);
      cur = ChunkPlan();
      cur.s = k;
      acc = 0;
    };
    for (size_t b = 0; b < nblk; ++b) {
      const size_t k0 = b * SDF_CUT_BLOCK, k1 = std::min(n, k0 + SDF_CUT_BLOCK);
      const BatchCut::Block &blk = cut.blocks[b];
      const size_t target = normal.empty() && nch_by_size > 1 ? first_target : chunk_target;
      // (all tasks of the range count towards the target, as they cost planning time whether they run or not)
      if (k0 > cur.s && k0 - cur.s >= target) close_at(k0);
      const uint64_t bbd = blk.bd - (hv ? blk.hbd : 0);  // (lane tasks are not in a block's sums)
      if (acc + bbd <= region_budget) {
        acc += bbd;
        cur.ntask += blk.nt - (hv ? blk.hnt : 0);
        cur.stage_words += blk.sw - (hv ? blk.hsw : 0);
        cur.order_cap += blk.oc - (hv ? blk.hoc : 0);
        continue;
      }
      for (size_t k = k0; k < k1; ++k) {
        if (hv && hv[k]) continue;
        if (ln && ln[k]) continue;
        const size_t bd = (size_t)bound[k] << 8;
        if (k > cur.s && acc + bd > region_budget) close_at(k);
        acc += bd;
        cur.ntask += cap[k] >> 31;
        cur.stage_words += cap[k] & 0x7fffffffu;
        cur.order_cap += (cap[k] >> 31) * order_entries(tasks[k]);
      }
    }
    cur.e = n;
    normal.push_back(cur);
    cut.region_need = std::max(cut.region_need, acc);
  }
  cut.region_need = (cut.region_need + 255) & ~(size_t)255;
  cut.heavy_need = (cut.heavy_need + 255) & ~(size_t)255;
  cut.nreg_ws = std::min(cut.max_regions, normal.size());
  cut.chunks = heavy_chunks;  // heavy first
  cut.chunks.insert(cut.chunks.end(), normal.begin(), normal.end());
  // bases: what the chunks before this one (in launch order) occupy
  size_t pb = 0, ob = 0;
  int64_t stage = 0;
  for (ChunkPlan &c : cut.chunks) {
    c.pb = pb;
    c.ob = ob;
    c.stage0 = stage;
    pb += c.ntask;
    ob += c.order_cap;
    stage += c.stage_words;
  }
  cut.ntask_total = pb;
  cut.order_total = ob;
  cut.stage_total = stage;
  if (dbg_cut && n >= 100000)
    fprintf(stderr, "[cut: clear + scan %.2f ms, boundaries %.2f ms]\n", std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count());
  return SDF_OK;
}

// Plans one chunk into plan[c.pb ...] and order[c.ob ...].
static void plan_chunk(const PlanEnv &env, const BatchCut &cut, ChunkPlan &c, PlanTask *plan, int32_t *order,
                       PlanScratch &sx) {
  using plan_detail::Cls;
  const sdf_task *tasks = env.tasks;
  const bool want_cigar = env.want_cigar;
  const bool dbg_pc = env.cfg->debug_plan != 0;
  const auto tp0 = std::chrono::steady_clock::now();
  auto tp1 = tp0, tp2 = tp0, tp3 = tp0;
  std::vector<int32_t> &win_need = sx.win_need, &partner = sx.partner;
  std::vector<Cls> &cls = sx.cls;
  win_need.clear();
  sx.bs_alt.clear();
  size_t np = c.pb;
  int64_t stage_words = c.stage0;
  size_t n_stripe_tasks = 0;   // stripe tasks of the chunk,
  double stripe_cells = 0;     // their cells,
  int stripe_rows = 0;         // the anti-diagonals of the longest of them
  // Strip kernels (extz2_strip.hip) are throughput kernels: a step of eight cells per lane is a microsecond on a wavefront
  // alone, a 500 x 500 task 0.55 ms where the window kernels take 0.3, a chain of twelve blocks 6.5 ms against the
  // stripe kernel's 3.0 (profiles/r03_strip.txt).  They take a chunk's tasks only when there are enough of them to fill
  // the device: 1,024 one-wavefront tasks, 3,072 chain wavefronts (1,024 for the heavy tasks of a large batch).
  bool use_strip = false, use_chain = false;
  if (env.strip_ok) {
    const bool force = env.strip_always;  // (tests: whatever the count)
    size_t n9 = 0, w10 = 0;
    for (size_t pos = c.s; pos < c.e; ++pos) {
      const size_t k = c.heavy ? cut.heavy_idx[pos] : pos;
      const sdf_task &t = c.heavy ? cut.heavy_tasks[pos] : tasks[k];
      if (t.tlen <= 256 || t.tlen > kStripChainMaxT || t.qlen < 64) continue;
      if (!c.heavy && cut.split_heavy && cut.heavy[k]) continue;
      if (t.w >= 0 && t.w < std::max(t.qlen, t.tlen)) continue;
      if (t.tlen <= kStripMaxT) ++n9;
      else w10 += (size_t)strip_blocks(t.tlen, 8);
    }
    use_strip = force || n9 >= 1024;
    // (the heavy tasks of a large batch share the device with everything else it holds: what counts there is the work per
    // cell, not the pace of a lone chain -- the hg19 mixture's 608 long tasks, 2,900 chain wavefronts: 13.5-14.0 ms as chains
    // against 14.2-15.2 on the stripe kernel; the same tasks alone: 10-15 % slower as chains)
    const size_t chain_min = c.heavy && cut.n_heavy * 16 <= env.n ? std::min<size_t>(env.chain_min, 1024) : env.chain_min;
    use_chain = force || w10 / 2 >= chain_min;
  }
  // A FEW tasks much longer than the rest of a chunk of chains: a chain's step is ~1 us on blocks of 256 columns, a stripe's row
  // 0.25 us, and the launch ends with its longest task -- up to 64 full-band tasks of 12,000 rows + columns and more take the
  // stripe kernel next to the chains (the chr1-sized stage run's far-gap round: eight tasks of 8.7 kb a side among 2,500 of
  // 1.7 kb: the round's DP 21 -> see profiles/r04_stage.txt).
  size_t n_long = 0;
  auto long_task = [&](const sdf_task &t) {
    return t.qlen + t.tlen >= 12000 && t.tlen > env.stripe_min && t.tlen <= kStripeMaxT && (t.w < 0 || t.w >= std::max(t.qlen, t.tlen)) &&
           stripe_lds_bytes(t.qlen, 4) <= (size_t)env.max_dyn_lds;
  };
  if (use_chain && !env.strip_always && !env.no_stripe) {
    for (size_t pos = c.s; pos < c.e; ++pos) {
      const size_t k = c.heavy ? cut.heavy_idx[pos] : pos;
      const sdf_task &t = c.heavy ? cut.heavy_tasks[pos] : tasks[k];
      if (!c.heavy && cut.split_heavy && cut.heavy[k]) continue;
      n_long += long_task(t);
    }
  }
  const bool long_to_stripes = n_long > 0 && n_long <= 64;
  for (size_t pos = c.s; pos < c.e; ++pos) {
    const size_t k = c.heavy ? cut.heavy_idx[pos] : pos;
    const sdf_task &t = c.heavy ? cut.heavy_tasks[pos] : tasks[k];
    if (!c.heavy && cut.split_heavy && cut.heavy[k]) continue;
    if (cut.use_lane && cut.lane[k]) continue;  // (planned on the device: extz2_lane.hip)
    if (!plan_detail::task_runs(t, env.degenerate)) continue;  // reference early return (:57,:81)
    PlanTask p;
    p.q_word = t.q_off;
    p.t_word = t.t_off;
    p.qlen = t.qlen;
    p.tlen = t.tlen;
    p.w = t.w < 0 ? std::max(t.qlen, t.tlen) : t.w;
    p.zdrop = t.zdrop;
    p.flag = t.flag | (want_cigar ? 0 : SDF_FLAG_SCORE_ONLY);
    int nc = std::min(t.qlen, t.tlen);
    nc = (std::min(nc, p.w + 1) + 15) / 16 + 1;
    p.ncol16 = nc * 16;
    p.out_idx = (int32_t)k;
    p.pad_ = 0;
    // register-resident wave kernel when only CIGAR/score/mte are wanted and the shape fits
    p.nreg = 0;
    int wneed = 0;
    bool plain_ok = false;
    {
      const int nrow = t.qlen + t.tlen - 1;
      Band bl;
      const bool band_whole = (p.w >= 1 || nrow == 1) && band_of(nrow - 1, t.qlen, t.tlen, p.w, bl);
      // (generic scoring and the approximate-max modes exist in the general kernel only)
      const int general_only = SDF_FLAG_RIGHT | SDF_FLAG_EXTZ_ONLY | SDF_FLAG_GENERIC_SC | SDF_FLAG_APPROX_MAX | SDF_FLAG_APPROX_DROP;
      const bool simple = !(env.want & SDF_WANT_EXT) && t.zdrop < 0 && !(t.flag & general_only) && env.gapo >= 0;
      const bool plain = simple && band_whole;
      plain_ok = plain && !env.force_general;
      // a long banded task, whether its band reaches the end or runs out: the banded stripe kernel (a chain of
      // qlen + tlen rows at a hundred instructions each instead of several hundred)
      // -- where the one-task kernels are slower per row: windows of more than 192 slots (six or eight registers on
      // one wavefront, or the general kernel: 0.9 - 1.9 us per row in a mixed batch against 0.2 - 0.5), and every band
      // that runs out (the TRACK flavour: 0.6 us per row with three registers, 1.35 with six; mm8-like mixture with
      // those of up to 192 slots left on it 27.5 ms, without 24.1)
      const int bstripe_all = (int)env.cfg->bstripe_all;  // (tests: every long banded task)
      constexpr int bstripe_plain_min = 192;  // (measured on the mm8-like mixture: 384 -> 47 ms, 192 -> 29-34 ms, 128 -> 36-39 ms)
      const int bneed = std::min(p.ncol16 + 32, (t.tlen + 15) / 16 * 16);
      if (simple && !env.force_general && env.bstripe_min_rows > 0 && nrow >= env.bstripe_min_rows && p.w >= 1 &&
          p.w < std::max(t.qlen, t.tlen) && (bstripe_all || bneed > bstripe_plain_min || !band_whole)) {
        const int nr = plan_detail::bstripe_nreg(t.tlen, (int)env.cfg->bstripe_nreg);
        if (nr && bstripe_lds_bytes(p.w, nr) <= (size_t)env.max_dyn_lds) {
          p.nreg = nr;
          p.pad_ = 7;
          if (plain_ok && bneed <= kMixedMaxNeed && !env.no_pair && !env.no_mixed) {  // (band_whole: a mixed pair can take it)
            const int wn = bneed <= 128 ? 1 : bneed <= 256 ? 2 : bneed <= 384 ? 3 : bneed <= 512 ? 4 : 6;
            if (wave_lds_bytes(t.qlen, t.tlen, wn) <= (size_t)env.max_dyn_lds) {
              sx.bs_alt.push_back((int32_t)(np - c.pb));
              sx.bs_alt.push_back(wn);
              sx.bs_alt.push_back(bneed);
            }
          }
        }
      }
      // the same request on a band that runs out before the end of both sequences: the pair kernel's TRACK flavour
      // (exact H of every cell, best cell for the traceback), the task paired with itself; windows up to 384 slots
      const bool runs_out = !band_whole && p.w >= 1 && nrow > 1;
      if (p.pad_ != 7 && runs_out && !env.force_general && !env.no_pair && !(env.want & SDF_WANT_EXT) && t.zdrop < 0 &&
          !(t.flag & general_only) && env.gapo >= 0) {
        const int need = std::min(p.ncol16 + 32, (t.tlen + 15) / 16 * 16);
        // (wider windows stay on the general kernel: ten registers on one wavefront were measured no faster per row
        // than its 1024-thread workgroup, 2.0 vs 1.75 us on 20 kb tasks at w = 512)
        const int nreg = need <= 192 ? 3 : need <= 384 ? 6 : 0;
        if (nreg && pair_lds_bytes(t.qlen, t.tlen, nreg) <= (size_t)env.max_dyn_lds) {
          p.nreg = nreg;
          p.pad_ = 6;  // (becomes layout 2 below; 6 marks the TRACK launch class until then)
        }
      }
      if (plain_ok && p.pad_ != 7) {
        // window slots: one 16-row block of slack below, the score refresh overshoot above -- but never
        // beyond the target's last 16-cell block (cells past it are not part of any window)
        const int need = std::min(p.ncol16 + 32, (t.tlen + 15) / 16 * 16);
        const int nreg = need <= 128 ? 1 : need <= 256 ? 2 : need <= 384 ? 3 : need <= 512 ? 4 : need <= 768 ? 6 : need <= 1024 ? 8 : 0;
        if (nreg && wave_lds_bytes(t.qlen, t.tlen, nreg) <= (size_t)env.max_dyn_lds) {
          p.nreg = nreg;
          wneed = need;
        }
      }
    }
    win_need.push_back(wneed);
    p.dir_off = 0;
    p.cig_cap = (p.flag & SDF_FLAG_SCORE_ONLY) ? 0 : t.qlen + t.tlen + 2;
    p.cig_slot = stage_words;
    stage_words += p.cig_cap;
    if (use_strip && p.pad_ != 7 && plain_ok && p.w >= std::max(t.qlen, t.tlen) && t.tlen > 256 && t.tlen <= kStripMaxT &&
        t.qlen >= 64 && t.qlen < (1 << 18)) {
      // full band, a few hundred target bases: row-major strips, two tasks per wavefront
      p.nreg = 8;  // (columns per lane)
      p.pad_ = 9;
      win_need.back() = 0;
    } else if ((use_chain || (env.strip_ok && t.tlen > kStripeMaxT)) && !(long_to_stripes && long_task(t)) && !env.no_stripe &&
               p.pad_ != 7 && plain_ok && p.w >= std::max(t.qlen, t.tlen) && t.tlen > kStripMaxT && t.tlen <= kStripChainMaxT &&
               t.qlen >= 64 && t.qlen < (1 << 18)) {
      // ... wider: the same strips, a wavefront per block of columns, chained through HBM.  Targets beyond the stripe
      // kernel's 32,512 bases are chains however few they are: the alternative is the workgroup kernel walking its window
      // through LDS at ~3 us per row (a 61,440 x 61,440 task alone: 2,070 ms there, 80 ms here)
      p.nreg = 8;  // (columns per lane: 8, or 4 when the chunk has few chains -- decided below)
      p.pad_ = 10;
      win_need.back() = 0;
    } else if (p.pad_ != 7 && plain_ok && !env.no_stripe && p.w >= std::max(t.qlen, t.tlen) &&
        t.tlen > env.stripe_min && t.tlen <= kStripeMaxT) {
      // wide full-band task: one wavefront per stripe of 128 * nreg target positions (nreg: after this pass)
      if (stripe_lds_bytes(t.qlen, 4) <= (size_t)env.max_dyn_lds) {
        p.nreg = 4;
        p.pad_ = 5;
        ++n_stripe_tasks;
        stripe_cells += (double)t.qlen * (double)t.tlen;
        stripe_rows = std::max(stripe_rows, t.qlen + t.tlen);
      }
    }
    if (!p.nreg) {
      // general kernel: state in LDS, or in an HBM scratch slab when it does not fit; the PLAIN flavour (packed
      // recurrence, H along the band edge only) when nothing but CIGAR / score / mte is wanted and the window
      // is wide enough for the 256- or 1024-thread instantiation
      const bool hbm = general_lds_bytes(t.qlen, t.tlen) > (size_t)env.max_dyn_lds;
      const int width = std::min(p.ncol16, (t.tlen + 15) / 16 * 16);
      if (plain_ok && !hbm && width > 256) p.pad_ = 3;
      else if (plain_ok && hbm && width > 1024) p.pad_ = 4;
      else p.pad_ = hbm ? 1 : 0;
    }
    plan[np++] = p;
  }
  tp1 = std::chrono::steady_clock::now();
  const size_t cnt = np - c.pb;
  c.cnt = cnt;
  c.nord = 0;
  c.launches.clear();
  c.layouts = 0;
  c.paired = 0;
  c.dir_bytes = 0;
  if (cnt == 0) return;
  PlanTask *cp = plan + c.pb;  // chunk-relative indexing below
  if (n_stripe_tasks) {
    // One stripe width per chunk (one launch, all its tasks side by side).  A wavefront alone on its SIMD issues an
    // instruction every ~5 cycles whatever its width, so a row of a 128-position stripe takes a quarter of the time of
    // a row of a 512-position one, and a task is a chain of qlen + tlen dependent rows: few tasks want narrow stripes
    // (a 6000 x 6000 task alone: 6.1 ms at 512 positions per stripe).  Many tasks want wide ones: per cell a wide stripe
    // spends fewer instructions on the edges and the loop (1k x 1k tasks by the thousand: 1180 Gcell/s at 512, 800 at
    // 128), and with a wavefront or more per SIMD the chains overlap anyway.
    const int force_nreg = (int)env.cfg->stripe_nreg;
    // -> the width with the smaller of: the launch's cells at the width's throughput (800 / 1100 / 1300 Gcell/s measured
    // on batches of equal tasks), its longest chain at the width's row time alone on a SIMD (0.24 / 0.27 / 0.50 us)
    int nr = 1;
    {
      const double rate[3] = {800e3, 1100e3, 1300e3}, row_us[3] = {0.24, 0.27, 0.50};  // cells per us; us per row
      double best = 1e300;
      for (int q = 0; q < 3; ++q) {
        const double t_us = std::max(stripe_cells / rate[q], (double)stripe_rows * row_us[q]);
        if (t_us < best) {
          best = t_us;
          nr = 1 << q;
        }
      }
      if (force_nreg) nr = force_nreg;
    }
    for (size_t k = 0; k < cnt; ++k) {
      PlanTask &p = cp[k];
      if (p.pad_ != 5) continue;
      p.nreg = nr;
      if ((p.tlen + 128 * nr - 1) / (128 * nr) > 254) {  // (entry encoding: 8 bits of stripe index, 255 = idle)
        p.nreg = 0;
        const bool hbm = general_lds_bytes(p.qlen, p.tlen) > (size_t)env.max_dyn_lds;
        p.pad_ = hbm ? 4 : 3;
      }
    }
  }

  // Mixed pairs are a throughput kernel (a wavefront of nine registers takes ~1.3 us per row when it shares its SIMD): the
  // chunk's banded tasks take them when there are enough to fill the device, and then the long ones whose band reaches the
  // end come back from the banded stripe kernel (0.73 VALU instructions per cell measured there against ~0.4-0.5 here,
  // profiles/r04_mm8_pmc.txt); in a chunk of few tasks a long task stays a chain of short stripe rows.
  bool use_mixed = false;
  std::vector<char> &mixedf = sx.mixedf;
  mixedf.assign(cnt, 0);
  if (!env.no_pair && !env.no_mixed && !env.force_general) {
    size_t ncand = sx.bs_alt.size() / 3;
    for (size_t k = 0; k < cnt; ++k)
      ncand += cp[k].nreg && cp[k].pad_ == 0 && win_need[k] > 0 && win_need[k] <= kMixedMaxNeed && cp[k].w < std::max(cp[k].qlen, cp[k].tlen);
    use_mixed = ncand >= env.mixed_min;
    if (use_mixed)
      for (size_t j = 0; j + 2 < sx.bs_alt.size(); j += 3) {
        PlanTask &p = cp[sx.bs_alt[j]];
        p.pad_ = 0;
        p.nreg = sx.bs_alt[j + 1];
        win_need[sx.bs_alt[j]] = sx.bs_alt[j + 2];
      }
  }
  // Pair kernel: two wave-eligible tasks of the chunk with the same (qlen, tlen, w, flag) and a window of at
  // most 512 slots share a wavefront (extz2_pair.hip).  partner[k] = the other task, or -1.  One pass with an
  // open-addressing table keyed by the geometry: entry = (first task seen with the key, the task of that key
  // still waiting for a partner or -1).
  partner.assign(cnt, -1);
  std::vector<char> &tracked = sx.tracked;
  tracked.assign(cnt, 0);
  for (size_t k = 0; k < cnt; ++k)
    if (cp[k].pad_ == 6) {
      cp[k].pad_ = 2;
      partner[k] = (int32_t)k;
      tracked[k] = 1;
    }
  if (!env.no_pair && !env.force_general) {
    auto &table = sx.table;
    size_t cap = 64;
    while (cap < 2 * cnt) cap *= 2;
    table.assign(cap, {-1, -1});
    for (size_t k = 0; k < cnt; ++k) {
      PlanTask &y = cp[k];
      if (!y.nreg || y.pad_ != 0 || win_need[k] > 512) continue;  // wave-kernel tasks only
      uint64_t h = ((uint64_t)(uint32_t)y.qlen * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)y.tlen * 0xC2B2AE3D27D4EB4Full) ^
                   ((uint64_t)(uint32_t)y.w * 0x165667B19E3779F9ull) ^ ((uint64_t)(uint32_t)y.flag << 40);
      h ^= h >> 29;
      for (size_t slot = (size_t)h & (cap - 1);; slot = (slot + 1) & (cap - 1)) {
        auto &e = table[slot];
        if (e.first < 0) {
          e = {(int32_t)k, (int32_t)k};
          break;
        }
        const PlanTask &x = cp[e.first];
        if (x.qlen != y.qlen || x.tlen != y.tlen || x.w != y.w || x.flag != y.flag) continue;
        if (e.second < 0) {
          e.second = (int32_t)k;
          break;
        }
        PlanTask &z = cp[e.second];
        const int regs = (win_need[k] + 63) / 64;
        const int nreg = regs <= 4 ? regs : regs <= 6 ? 6 : 8;
        if (pair_lds_bytes(y.qlen, y.tlen, nreg) > (size_t)env.max_dyn_lds) break;
        z.nreg = y.nreg = nreg;
        z.pad_ = y.pad_ = 2;
        partner[k] = e.second;
        partner[e.second] = (int32_t)k;
        c.paired += 2;
        e.second = -1;
        break;
      }
    }
    if (use_mixed) {
      // Mixed pairs: the tasks without a partner of their geometry (and those whose window of 513..576 slots no other pair
      // kernel holds), sorted by (w, flag, last clip-free row): neighbours share the most rows.  Worth it when the shared
      // rows are at least half of the longer task's (a row of a pair costs ~1.3 rows of the one-task wave kernel).
      std::vector<uint64_t> &keys = sx.mix_keys;
      keys.clear();
      auto add_key = [&](size_t k) {
        const PlanTask &y = cp[k];
        if (y.w >= std::max(y.qlen, y.tlen) || y.w >= (1 << 14) || (y.flag & ~0xff) || k >= ((size_t)1 << 22)) return;
        int cf = pair_clip_free(y.qlen, y.tlen, y.w);
        cf = cf < 0 ? 0 : cf > 0xfffff ? 0xfffff : cf;
        keys.push_back(((uint64_t)(uint32_t)y.w << 50) | ((uint64_t)(uint32_t)(y.flag & 0xff) << 42) | ((uint64_t)cf << 22) | (uint64_t)k);
      };
      for (auto &e : table)
        if (e.second >= 0) add_key((size_t)e.second);
      for (size_t k = 0; k < cnt; ++k)
        if (cp[k].nreg && cp[k].pad_ == 0 && win_need[k] > 512 && win_need[k] <= kMixedMaxNeed) add_key(k);
      // (a batch of few geometries -- the headline's 100,000 tasks of ~50 target lengths -- leaves a handful without a
      // partner: they are paired with themselves inside the tuned launch, as before, not given a launch of their own)
      if (keys.size() * 16 < cnt) keys.clear();
      std::sort(keys.begin(), keys.end());
      for (size_t q = 0; q + 1 < keys.size();) {
        const size_t a = (size_t)(keys[q] & 0x3fffffu), b = (size_t)(keys[q + 1] & 0x3fffffu);
        PlanTask &x = cp[a], &y = cp[b];
        bool ok = (keys[q] >> 42) == (keys[q + 1] >> 42);  // same band and flags
        int nreg = 0;
        if (ok) {
          const int shared = (x.qlen == y.qlen && x.tlen == y.tlen) ? x.qlen + x.tlen : pair_shared_rows(x.qlen, x.tlen, y.qlen, y.tlen, x.w);
          ok = 2 * shared >= std::max(x.qlen + x.tlen, y.qlen + y.tlen);
          const int regs = (std::max(win_need[a], win_need[b]) + 63) / 64;
          nreg = regs <= 2 ? 2 : regs <= 6 ? regs : regs <= 8 ? 8 : 9;
          ok = ok && pair_mixed_lds_bytes(std::max(x.qlen, y.qlen), std::max(x.tlen, y.tlen), nreg) <= (size_t)env.max_dyn_lds;
        }
        if (!ok) {
          ++q;
          continue;
        }
        x.nreg = y.nreg = nreg;
        x.pad_ = y.pad_ = 2;
        partner[a] = (int32_t)b;
        partner[b] = (int32_t)a;
        mixedf[a] = mixedf[b] = 1;
        c.paired += 2;
        q += 2;
      }
      // ... and what is left of them: a long task alone would be a launch of its own on the one-task kernels, as long as its
      // chain of rows; paired with itself it is one more wavefront of the mixed launch (a short one keeps its old routes)
      size_t n_mixed = 0;
      for (size_t q = 0; q < keys.size(); ++q) {
        const size_t a = (size_t)(keys[q] & 0x3fffffu);
        if (partner[a] >= 0) {
          ++n_mixed;
          continue;
        }
        PlanTask &x = cp[a];
        if (x.qlen + x.tlen < 2000) continue;
        const int regs = (win_need[a] + 63) / 64;
        const int nreg = regs <= 2 ? 2 : regs <= 6 ? regs : regs <= 8 ? 8 : 9;
        if (pair_mixed_lds_bytes(x.qlen, x.tlen, nreg) > (size_t)env.max_dyn_lds) continue;
        x.nreg = nreg;
        x.pad_ = 2;
        partner[a] = (int32_t)a;
        mixedf[a] = 1;
      }
      // pairs of equal geometry among the banded tasks join the mixed launches when they are the few (a batch of tasks of all
      // lengths: one launch fewer per register count, each as long as its longest chain); a batch of few geometries keeps
      // its tuned instantiations
      size_t n_exact = 0;
      for (size_t k = 0; k < cnt; ++k)
        n_exact += cp[k].pad_ == 2 && !mixedf[k] && !tracked[k] && partner[k] >= 0 && partner[k] != (int32_t)k && cp[k].w < std::max(cp[k].qlen, cp[k].tlen);
      if (n_exact * 4 <= n_mixed)
        for (size_t k = 0; k < cnt; ++k) {
          PlanTask &x = cp[k];
          if (x.pad_ != 2 || mixedf[k] || tracked[k] || partner[k] <= (int32_t)k || x.w >= std::max(x.qlen, x.tlen)) continue;
          const int regs = (win_need[k] + 63) / 64;
          const int nreg = regs <= 2 ? 2 : regs <= 6 ? regs : regs <= 8 ? 8 : 9;
          if (pair_mixed_lds_bytes(x.qlen, x.tlen, nreg) > (size_t)env.max_dyn_lds) continue;
          x.nreg = cp[partner[k]].nreg = nreg;
          mixedf[k] = mixedf[partner[k]] = 1;
        }
      for (auto &e : table)  // (no longer waiting)
        if (e.second >= 0 && partner[e.second] >= 0) e.second = -1;
      // a long task that came over from the banded stripe kernel and found nobody goes back to it
      for (size_t j = 0; j + 2 < sx.bs_alt.size(); j += 3) {
        const int32_t k = sx.bs_alt[j];
        if (partner[k] >= 0) continue;
        cp[k].pad_ = 7;
        cp[k].nreg = plan_detail::bstripe_nreg(cp[k].tlen, (int)env.cfg->bstripe_nreg);
        win_need[k] = 0;
        for (auto &e : table)
          if (e.second == k) e.second = -1;
      }
    }
    // a task left without a partner is paired with itself (both halves compute the same task and write the
    // same bytes) instead of occupying a launch of its own for a whole task latency -- when they are few.  Many of them
    // (a batch of banded tasks of all lengths: BASELINE configs[4]) fill launches of their own on the one-task wave
    // kernel, whose row is 30-40 % shorter than that of a wavefront whose halves compute the same thing.
    size_t n_left = 0;
    for (auto &e : table) n_left += e.second >= 0;
    for (auto &e : table) {
      if (e.second < 0 || n_left > env.self_pair_max) continue;
      PlanTask &y = cp[e.second];
      const int regs = (win_need[e.second] + 63) / 64;
      const int nreg = regs <= 4 ? regs : regs <= 6 ? 6 : 8;
      if (pair_lds_bytes(y.qlen, y.tlen, nreg) > (size_t)env.max_dyn_lds) continue;
      y.nreg = nreg;
      y.pad_ = 2;
      partner[e.second] = e.second;
    }
  }

  {  // strip kernel: two tasks per wavefront, neighbours in the order of (column blocks, rows, columns) -- the wavefront
     // steps through the larger of its two matrices; a task left over is paired with itself
    std::vector<int32_t> &sl = sx.stripe_lane;  // (scratch)
    std::vector<uint64_t> &keys = sx.strip_keys;
    keys.clear();
    // Chains: a wavefront alone on its SIMD issues an instruction every ~7 cycles whatever it does, a chain of blocks runs
    // at the pace of its steps, and a step of four columns is half as long as one of eight: few chains (the few long
    // tasks of a batch: fewer wavefronts than the device has slots for) take four columns per lane, many take eight
    // (less per-step overhead per cell).  One width per chunk: one launch.
    {
      const int force_cols = env.strip_cols;
      size_t chain_waves = 0;
      for (size_t k = 0; k < cnt; ++k)
        if (cp[k].pad_ == 10) chain_waves += (size_t)strip_blocks(cp[k].tlen, 8);
      // (measured, profiles/r03_strip.txt: four columns shorten a lone chain -- 5.3 against 6.5 ms for 6000 x 6000 -- but
      // cost throughput -- 1,364 against 1,612 Gcell/s on 20,000 x 1000 x 1000 --, and lone chains are the stripe
      // kernel's anyway: eight unless SDF_STRIP_COLS says otherwise)
      // (round 4: a chunk whose chains leave the SIMDs with fewer than four of their wavefronts each takes four -- the heavy
      // chunk of the hg19 mixture, 608 long tasks = 2,900 wavefronts of eight columns: 14.0-14.2 against 14.65-14.85 ms for the
      // 1,000,000-task batch, its DP 9.3-10.8 against 11.6-11.9 ms)
      const int cols = force_cols == 4 || force_cols == 8 ? force_cols : chain_waves / 2 <= 4096 ? 4 : 8;  // (two tasks a wavefront)
      for (size_t k = 0; k < cnt; ++k)
        if (cp[k].pad_ == 10) cp[k].nreg = cols;
    }
    for (size_t k = 0; k < cnt; ++k)
      if (cp[k].pad_ == 9 || cp[k].pad_ == 10) {
        // (blocks, rows, columns within the last block, index): 8 + 18 + 9 + 24 bits -- up to 256 blocks; a one-wavefront
        // task has one block and a chain at least two, so the two kinds stay apart
        const int bw = 64 * cp[k].nreg;
        keys.push_back(((uint64_t)(strip_blocks(cp[k].tlen, cp[k].nreg) - 1) << 51) | ((uint64_t)(uint32_t)cp[k].qlen << 33) |
                       ((uint64_t)(uint32_t)((cp[k].tlen - 1) % bw) << 24) | (uint64_t)k);
      }
    // (8,747 keys in the heavy chunk of the hg19 mixture, planned in front of the call's first launch: a byte-wise radix
    // sort over the bytes that differ takes 0.05 ms where std::sort takes 0.25)
    if (keys.size() >= 2048) {
      std::vector<uint64_t> &tmp = sx.strip_keys_tmp;
      tmp.resize(keys.size());
      uint64_t all_or = 0, all_and = ~0ull;
      for (uint64_t kx : keys) {
        all_or |= kx;
        all_and &= kx;
      }
      const uint64_t varying = all_or ^ all_and;
      for (int sh = 0; sh < 64; sh += 8) {
        if (((varying >> sh) & 0xffu) == 0) continue;  // every key has the same byte here
        size_t cnt8[257] = {0};
        for (uint64_t kx : keys) ++cnt8[((kx >> sh) & 0xffu) + 1];
        for (int b = 0; b < 256; ++b) cnt8[b + 1] += cnt8[b];
        for (uint64_t kx : keys) tmp[cnt8[(kx >> sh) & 0xffu]++] = kx;
        keys.swap(tmp);
      }
    } else {
      std::sort(keys.begin(), keys.end());
    }
    sl.resize(keys.size());
    for (size_t q = 0; q < keys.size(); ++q) sl[q] = (int32_t)(keys[q] & 0xffffffu);
    for (size_t q = 0; q < sl.size();) {
      const int32_t x = sl[q];
      int32_t y = x;  // the next one, if it has as many column blocks and at most an eighth as many rows again (+ 64: the
                      // flag bound of cut_batch counts on it)
      if (q + 1 < sl.size() && cp[sl[q + 1]].pad_ == cp[x].pad_ &&
          strip_blocks(cp[sl[q + 1]].tlen, cp[x].nreg) == strip_blocks(cp[x].tlen, cp[x].nreg) &&
          cp[sl[q + 1]].qlen <= cp[x].qlen + cp[x].qlen / 8 + 64)
        y = sl[q + 1];
      q += y == x ? 1 : 2;
      const int32_t lo = std::min(x, y), hi = std::max(x, y);
      partner[lo] = hi;
      partner[hi] = lo;
      // (the rows the wavefront steps through: the traceback finds a block's records by it)
      cp[lo].ncol16 = cp[hi].ncol16 = std::max(cp[lo].qlen, cp[hi].qlen) | (lo == hi ? 1 << 30 : 0);  // (bit 30: no partner)
      if (cp[lo].pad_ == 10) {  // (the chain kernel finds a task's partner through its plan record)
        cp[lo].zdrop = hi;
        cp[hi].zdrop = lo;
      }
      if (lo != hi) c.paired += 2;
    }
  }
  tp2 = std::chrono::steady_clock::now();
  // launch classes: (kernel, LDS bytes rounded to a power of two); the direction-flag layout inside this chunk's
  // workspace region is fixed in the same pass
  cls.clear();
  size_t dir_acc = 0;
  for (size_t k = 0; k < cnt; ++k) {
    PlanTask &p = cp[k];
    {
      size_t need = 0;
      // (the stripe kernels keep their inter-stripe words behind the flag blocks: reserved for score-only tasks as well)
      if (p.pad_ == 9 || p.pad_ == 10) {
        // one flag region per wavefront: task A's words are the even, its partner's the odd ones of the same records
        if (partner[k] < (int32_t)k) {
          need = 0;  // (placed with its partner, below)
        } else {
          const PlanTask &pb = cp[partner[k]];
          const int qm = std::max(p.qlen, pb.qlen), tm = std::max(p.tlen, pb.tlen);
          need = (strip_dir_bytes(qm, tm, p.nreg, partner[k] == (int32_t)k) + 255) & ~(size_t)255;
          if (p.pad_ == 10) need += strip_chain_sync_bytes(qm, tm, p.nreg);
          if (partner[k] != (int32_t)k) cp[partner[k]].dir_off = (int64_t)dir_acc + 4;
        }
      } else if (p.pad_ == 5)
        need = (stripe_dir_bytes(p.qlen, p.tlen, p.nreg) + stripe_sync_bytes(p.qlen, p.tlen, p.nreg) + 255) & ~(size_t)255;
      else if (p.pad_ == 7)
        need = (bstripe_dir_bytes(p.qlen, p.tlen, p.w, p.nreg) + bstripe_sync_bytes(p.qlen, p.tlen, p.w, p.nreg) + 255) & ~(size_t)255;
      else if (!(p.flag & SDF_FLAG_SCORE_ONLY)) {
        const size_t nblk = (size_t)((p.qlen + p.tlen - 1 + 15) / 16);
        if (p.pad_ == 2) need = nblk * (size_t)p.nreg * 512;
        else if (p.nreg) need = nblk * (size_t)p.nreg * 1024;
        else need = ((size_t)(p.qlen + p.tlen - 1) * (size_t)p.ncol16 + 16 + 255) & ~(size_t)255;
      }
      if (!((p.pad_ == 9 || p.pad_ == 10) && partner[k] < (int32_t)k)) p.dir_off = (int64_t)dir_acc;
      dir_acc += need;
    }
    c.layouts |= 1u << (p.nreg == 0 ? 0 : p.pad_ == 2 ? 2 : p.pad_ == 5 ? 3 : p.pad_ == 7 ? 4 : (p.pad_ == 9 || p.pad_ == 10) ? 6 : 1);
    const int width = std::min(p.ncol16, (p.tlen + 15) / 16 * 16);
    int bs = width > 1024 ? 1024 : width > 256 ? 256 : 64;  // 4 cells per thread and pass over the row
    size_t lds = 2048, need;
    if (p.pad_ == 10) {
      if (partner[k] < (int32_t)k) continue;  // (its partner's entry stands for both)
      bs = 600 + p.nreg;  // chained strips: one wavefront (workgroup) per block of 64 x nreg columns of a pair of tasks
      need = 16;
      lds = 1024;
    } else if (p.pad_ == 9) {
      if (partner[k] < (int32_t)k) continue;  // placed together with its partner
      bs = 500;  // strip kernel, one wavefront per pair of tasks
      need = strip_lds_bytes(std::max(p.qlen, cp[partner[k]].qlen), std::max(p.tlen, cp[partner[k]].tlen));
      lds = 1024;
      while (lds < need) lds *= 2;
    } else if (p.pad_ == 2) {
      if (partner[k] < (int32_t)k) continue;  // placed together with its partner
      // 100 + NREG; + 10 for the streamed-window instantiation (sequences longer than the LDS windows)
      bs = tracked[k] ? 120 + p.nreg : mixedf[k] ? 130 + p.nreg : 100 + p.nreg + (pair_fits_whole(p.qlen, p.tlen, p.nreg) ? 0 : 10);
      need = mixedf[k] ? pair_mixed_lds_bytes(std::max(p.qlen, cp[partner[k]].qlen), std::max(p.tlen, cp[partner[k]].tlen), p.nreg)
                       : pair_lds_bytes(p.qlen, p.tlen, p.nreg);
      lds = 8192;
      while (lds < need) lds *= 2;
    } else if (p.pad_ == 5) {
      bs = 300 + p.nreg;  // stripe kernel, one wavefront (workgroup) per stripe
      need = stripe_lds_bytes(p.qlen, p.nreg);
      lds = 8192;
      while (lds < need) lds *= 2;
    } else if (p.pad_ == 7) {
      bs = 400 + p.nreg;  // banded stripe kernel, one wavefront (workgroup) per stripe
      need = bstripe_lds_bytes(p.w, p.nreg);
      lds = 8192;
      while (lds < need) lds *= 2;
    } else if (p.nreg) {
      // NREG; + 10 for the streamed-window instantiation (sequences longer than the LDS windows)
      bs = p.nreg + (wave_fits_whole(p.qlen, p.tlen, p.nreg) ? 0 : 10);
      // one class for everything up to 6 KiB (>= 6 waves/SIMD either way), powers of two above
      need = wave_lds_bytes(p.qlen, p.tlen, p.nreg);
      lds = 6144;
      while (lds < need) lds *= 2;
    } else if (p.pad_ == 1 || p.pad_ == 4) {  // HBM-resident state: one class, slab = largest requirement
      bs = p.pad_ == 4 ? 2001 : width > 1024 ? 1001 : 1000;
      need = general_lds_bytes(p.qlen, p.tlen);
      lds = (size_t)1 << 40;
    } else {
      if (p.pad_ == 3) bs += 2000;  // PLAIN flavour: 2256 / 3024
      need = general_lds_bytes(p.qlen, p.tlen);
      while (lds < need) lds *= 2;
    }
    const bool hbm_cls = bs == 1000 || bs == 1001 || bs == 2001;
    if (!hbm_cls && need > (size_t)env.max_dyn_lds) {  // every kernel choice above checked its LDS need
      c.err = "internal: launch class needs more LDS than the device offers";
      return;
    }
    if (!hbm_cls && lds > (size_t)env.max_dyn_lds) lds = env.max_dyn_lds;
    Cls *cl = nullptr;
    for (auto &x : cls)
      if (x.bs == bs && x.lds == lds) cl = &x;
    if (!cl) {
      cls.push_back({bs, lds, 0, {}});
      cl = &cls.back();
    }
    cl->need_max = std::max(cl->need_max, need);
    {  // rough per-workgroup rates: general 64 / 256 / 1024 threads, HBM state, wave, pair
      const double rate = bs == 64 ? 0.03 : bs == 256 ? 0.1 : bs == 1024 ? 0.6 : bs == 2256 ? 0.3 : bs == 3024 ? 0.85
                          : bs == 2001 ? 0.3 : bs >= 1000 ? 0.08 : bs >= 600 ? 2.0 : bs >= 500 ? 0.5 : bs >= 400 ? 0.3 : bs >= 300 ? 1.0 : bs >= 200 ? 2.2 : bs >= 100 ? 0.25 : 0.13;  // (pair classes are 100..126)
      cl->est = std::max(cl->est, (double)(p.qlen + p.tlen) * (double)p.ncol16 / rate);
    }
    cl->idx.push_back((int32_t)k);
    if (p.pad_ == 2 || p.pad_ == 9) cl->idx.push_back(partner[k]);
  }
  c.dir_bytes = dir_acc;
  if (dir_acc > (c.heavy ? cut.heavy_need : cut.region_need)) {
    c.err = "internal: direction-flag region overflow";
    return;
  }
  // Small classes of one kernel (< 2048 tasks: occupancy is not what limits them, their longest task is) are
  // merged into one launch with the largest LDS size among them: fewer launches queued one behind the other.
  for (size_t a = 0; a < cls.size(); ++a) {
    if (cls[a].idx.empty() || cls[a].idx.size() >= 2048) continue;
    for (size_t b = a + 1; b < cls.size(); ++b) {
      if (cls[b].bs != cls[a].bs || cls[b].idx.empty() || cls[b].idx.size() >= 2048) continue;
      cls[a].lds = std::max(cls[a].lds, cls[b].lds);
      cls[a].need_max = std::max(cls[a].need_max, cls[b].need_max);
      cls[a].est = std::max(cls[a].est, cls[b].est);
      cls[a].idx.insert(cls[a].idx.end(), cls[b].idx.begin(), cls[b].idx.end());
      cls[b].idx.clear();
    }
  }
  cls.erase(std::remove_if(cls.begin(), cls.end(), [](const Cls &x) { return x.idx.empty(); }), cls.end());
  // inside a launch of few tasks the longest go first too (workgroups are dispatched in order: a long task that
  // starts last is the tail of the launch); pair-kernel entries move as (task, partner) units
  for (auto &x : cls) {
    const bool mixed_cls = x.bs >= 130 && x.bs < 140;  // (long and short tasks in one launch: the long chains start first)
    if ((x.idx.size() >= 8192 && !mixed_cls) || x.idx.size() < 3) continue;
    auto work = [&](int32_t k) { return (int64_t)(cp[k].qlen + cp[k].tlen) * cp[k].ncol16; };
    int64_t wmin = work(x.idx[0]), wmax = wmin;
    for (int32_t k : x.idx) {
      const int64_t wk = work(k);
      wmin = std::min(wmin, wk);
      wmax = std::max(wmax, wk);
    }
    if (wmax < 2 * wmin) continue;  // tasks of one size: the order does not matter
    if ((x.bs >= 100 && x.bs < 200) || x.bs == 500) {
      std::vector<std::pair<int32_t, int32_t>> pr(x.idx.size() / 2);
      for (size_t q = 0; q < pr.size(); ++q) pr[q] = {x.idx[2 * q], x.idx[2 * q + 1]};
      std::stable_sort(pr.begin(), pr.end(), [&](const std::pair<int32_t, int32_t> &a, const std::pair<int32_t, int32_t> &b) {
        return work(a.first) > work(b.first);
      });
      for (size_t q = 0; q < pr.size(); ++q) {
        x.idx[2 * q] = pr[q].first;
        x.idx[2 * q + 1] = pr[q].second;
      }
    } else {
      std::stable_sort(x.idx.begin(), x.idx.end(), [&](int32_t a, int32_t b) { return work(a) > work(b); });
    }
  }
  tp3 = std::chrono::steady_clock::now();
  // longest launches first so the long tasks start early
  std::sort(cls.begin(), cls.end(), [](const Cls &a, const Cls &b) { return a.est > b.est; });
  size_t cursor = 0;
  for (auto &x : cls) {
    const bool hbm_cls = x.bs == 1000 || x.bs == 1001 || x.bs == 2001;
    const size_t lds_bytes = hbm_cls ? ((x.need_max + 255) & ~(size_t)255) : std::min(x.lds, (x.need_max + 511) & ~(size_t)511);
    if ((x.bs >= 300 && x.bs < 500) || x.bs == 604 || x.bs == 608) {
      // Stripe kernel: one entry per stripe, (stripe << 24) | task.  Workgroup i runs on XCD i mod 8 and workgroups
      // are dispatched in index order.  The tasks are dealt to the eight residues (most stripes first, to the residue
      // with the fewest so far): a task's stripes share an XCD (its L2 carries their edge words).  A task is a chain:
      // stripe s starts 128 * nreg rows after stripe s - 1 and the last one ends qlen + tlen rows after the first
      // began, and the launch ends with its longest chain.  Every residue lists its entries by the row at which they
      // would have to start for all chains to END together -- s * 128 * nreg - (qlen + tlen) -- earliest first: the
      // long tasks get their wavefront slots first (those of their later stripes sleep until their turn), the short
      // ones fill in behind.  The key grows with s, so a stripe's left neighbour -- the only wavefront it ever waits
      // for -- has a smaller index on the same XCD: resident or finished.
      // (Entries with stripe index 255 do nothing: they keep the residues aligned where the lists differ in length.)
      // (chained strips, extz2_strip.hip: an entry stands for a PAIR of tasks -- the one listed and its partner --, a
      // "stripe" is a block of 512 columns of the wider of the two, 64 steps behind the block to its left, and the chain
      // is as long as the rows of the longer plus 64 per block)
      const bool chained = x.bs >= 600;
      const bool banded = !chained && x.bs >= 400;  // (banded stripe kernel: stripes over the padded target, 2 * 128 * nreg rows apart)
      const int nreg = chained ? x.bs - 600 : x.bs - (banded ? 400 : 300);
      const size_t first = cursor;
      const int nslot = chained ? 64 : banded ? 256 * nreg : 128 * nreg;  // rows between the starts of consecutive stripes
      auto stripes_of = [&](int32_t rel) {
        if (chained) return strip_blocks(std::max(cp[rel].tlen, cp[partner[rel]].tlen), nreg);
        return banded ? bstripe_geom(cp[rel].qlen, cp[rel].tlen, cp[rel].w, nreg).nst : (cp[rel].tlen + 128 * nreg - 1) / (128 * nreg);
      };
      auto chain_rows = [&](int32_t rel) {
        if (chained) return std::max(cp[rel].qlen, cp[partner[rel]].qlen) + 64 * stripes_of(rel);
        return cp[rel].qlen + cp[rel].tlen;
      };
      // (the start keys are taken in units of the rows between two stripes' starts: a counting sort per residue)
      int rq_max = 0, lane_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int32_t rel : x.idx) rq_max = std::max(rq_max, chain_rows(rel) / nslot);
      const int nbucket = rq_max + 258;  // key of (task, s): s - (qlen + tlen) / nslot + rq_max, 0 <= s < 255
      std::vector<int32_t> &lane_of = sx.stripe_lane, &fill = sx.stripe_fill;
      lane_of.resize(x.idx.size());
      fill.assign((size_t)8 * nbucket + 1, 0);
      for (size_t j = 0; j < x.idx.size(); ++j) {
        const int32_t rel = x.idx[j];
        int to = 0;
        for (int q = 1; q < 8; ++q)
          if (lane_sum[q] < lane_sum[to]) to = q;
        const int nst = stripes_of(rel), key0 = rq_max - chain_rows(rel) / nslot;
        lane_of[j] = to;
        lane_sum[to] += nst;
        for (int sidx = 0; sidx < nst; ++sidx) ++fill[(size_t)to * nbucket + key0 + sidx + 1];
      }
      const int longest = *std::max_element(lane_sum, lane_sum + 8);
      if (cursor + (size_t)longest * 8 > c.order_cap) {
        c.err = "internal: launch-order segment overflow";
        return;
      }
      for (size_t q = 0; q < (size_t)8 * nbucket; ++q) fill[q + 1] += fill[q];  // -> first position of (residue, key)
      {
        int base[8], acc = 0;  // positions are residue-relative
        for (int q = 0; q < 8; ++q) {
          base[q] = acc;
          acc += lane_sum[q];
        }
        const int32_t idle = (int32_t)((255u << 24) | (uint32_t)x.idx[0]);
        for (int pos = 0; pos < longest; ++pos)
          for (int q = 0; q < 8; ++q)
            if (pos >= lane_sum[q]) order[c.ob + first + (size_t)pos * 8 + q] = idle;
        for (size_t j = 0; j < x.idx.size(); ++j) {
          const int32_t rel = x.idx[j];
          const int to = lane_of[j], nst = stripes_of(rel), key0 = rq_max - chain_rows(rel) / nslot;
          for (int sidx = 0; sidx < nst; ++sidx) {
            const int pos = fill[(size_t)to * nbucket + key0 + sidx]++ - base[to];
            order[c.ob + first + (size_t)pos * 8 + to] = (int32_t)(((uint32_t)sidx << 24) | (uint32_t)rel);
          }
        }
        cursor = first + (size_t)longest * 8;
      }
      c.launches.push_back({x.bs, lds_bytes, first, cursor - first, x.est});
      for (int32_t rel : x.idx) c.launches.back().rmax = std::max(c.launches.back().rmax, cp[rel].qlen + cp[rel].tlen);
      continue;
    }
    c.launches.push_back({x.bs, lds_bytes, cursor, x.idx.size(), x.est});
    std::copy(x.idx.begin(), x.idx.end(), order + c.ob + cursor);
    cursor += x.idx.size();
  }
  c.nord = cursor;
  if (c.nord > c.order_cap) c.err = "internal: launch-order segment overflow";
  const bool dbg_pc_all = env.cfg->debug_plan >= 2;
  if (dbg_pc && (c.heavy || dbg_pc_all)) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[plan_chunk %s %zu tasks: tasks %.2f ms, pairing %.2f ms, classes %.2f ms, order %.2f ms]\n", c.heavy ? "heavy" : "ordinary", c.cnt, ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, std::chrono::steady_clock::now()));
  }
}

}  // namespace sdf
