// `sedef align bucket`: seed hits -> N bucket files for `align generate` (what bucket_alignments_extern of the reference
// produces, src/align_main.cc:38-198, with merge() of src/merge.cc:35-109 and the chromosome grouping of
// src/search_main.cc:93-120).
//
// The reference spills the extended seeds into one temporary file per pair of chromosome groups, re-reads each file to
// merge it, writes it back, and re-reads everything a third time to deal the hits into the buckets.  Here the seeds
// stream through memory once: they are grouped under the temporary file's NAME (its lexicographic order is the order the
// reference processes the groups in), every group is merged in place, and the hits are dealt straight into the bucket
// files.  A hit still passes through its BED text between the steps, exactly where the reference writes and re-reads it
// (the round trip drops fields: a seed's 14th column becomes the jaccard count, the CIGAR column is ignored).
#include <glob.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <tuple>

#include "sedef_host.h"

namespace sdfh {

namespace {
// a hit is kept with the pair ordered by (chromosome, start, end) of its two sides (src/merge.cc:38-46,
// src/align_main.cc:79-84); only the names, not the Sequence objects, change sides
void order_sides(Hit &h) {
  if (std::tie(h.query->name, h.query_start, h.query_end) > std::tie(h.ref->name, h.ref_start, h.ref_end)) {
    std::swap(h.query->name, h.ref->name);
    std::swap(h.query_start, h.ref_start);
    std::swap(h.query_end, h.ref_end);
  }
}

int complexity_class(const Hit &h) {  // src/align_main.cc:128-130, :172-174
  return (int)std::sqrt(double(h.query_end - h.query_start) * double(h.ref_end - h.ref_start)) / 1000;
}
int complexity_of(const Hit &h) {
  return (int)std::sqrt(double(h.query_end - h.query_start) * double(h.ref_end - h.ref_start));
}
}  // namespace

// Sweep over the hits in (strand, query chromosome, reference chromosome, query start, reference start) order.  The
// hits of the current run of nearby query intervals are kept as "windows" keyed by their reference end; a new hit
// swallows every window it comes within `merge_dist` of (in both sequences), repeatedly, as its own extent grows.
std::vector<Hit> merge_hits(std::vector<Hit> &hits, int merge_dist) {
  for (auto &h : hits) order_sides(h);
  // (std::sort, this comparator, this input order: equal keys end up where the reference's sort leaves them)
  std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) {
    return std::tie(a.ref->is_rc, a.query->name, a.ref->name, a.query_start, a.ref_start) <
           std::tie(b.ref->is_rc, b.query->name, b.ref->name, b.query_start, b.ref_start);
  });
  std::vector<Hit> merged;
  std::multimap<int, Hit> windows;  // by reference end
  auto flush = [&] {
    for (auto &w : windows) merged.push_back(w.second);
    windows.clear();
  };
  auto apart = [&](const Hit &w, const Hit &h) {
    return w.query_end + merge_dist < h.query_start || w.ref_end < h.ref_start - merge_dist ||
           w.ref_start > h.ref_end + merge_dist;
  };
  Hit last;  // the previous hit, its query end raised to that of everything before it in the run (src/merge.cc:101-102)
  bool have_last = false;
  for (Hit &h : hits) {
    const bool self = h.query->name == h.ref->name && h.query_start == h.ref_start && h.query_end == h.ref_end &&
                      h.query->is_rc == h.ref->is_rc;
    if (self) continue;  // a region against itself
    const bool new_run = !have_last || last.query_end + merge_dist < h.query_start ||
                         last.query->name != h.query->name || last.ref->name != h.ref->name ||
                         last.ref->is_rc != h.ref->is_rc;
    if (new_run) {
      flush();
    } else {
      for (bool grew = true; grew;) {
        grew = false;
        for (auto w = windows.lower_bound(h.ref_start - merge_dist); w != windows.end();) {
          if (apart(w->second, h)) {
            ++w;
            continue;
          }
          h.query_start = std::min(h.query_start, w->second.query_start);
          h.query_end = std::max(h.query_end, w->second.query_end);
          h.ref_start = std::min(h.ref_start, w->second.ref_start);
          h.ref_end = std::max(h.ref_end, w->second.ref_end);
          w = windows.erase(w);
          grew = true;
        }
      }
    }
    windows.emplace(h.ref_end, h);
    if (!new_run) h.query_end = std::max(h.query_end, last.query_end);
    last = h;
    have_last = true;
  }
  flush();
  return merged;
}

// chromosomes by (length, name) descending, packed greedily into groups of at most 100 MB
std::vector<std::vector<std::string>> generate_translation(const std::string &ref_path) {  // src/search_main.cc:93-120
  std::ifstream fai((ref_path + ".fai").c_str());
  if (!fai.is_open()) throw "Index file " + ref_path + ".fai does not exist";
  std::map<std::string, std::pair<size_t, std::string>> by_key;  // first token of the name -> (length, full name)
  for (std::string line; std::getline(fai, line);) {
    const auto f = split(line, '\t');
    if (f.size() != 5) throw "Index file " + ref_path + ".fai is malformed";
    by_key.insert({split(f[0], ' ').at(0), {(size_t)atoi(f[1].c_str()), f[0]}});
  }
  std::vector<std::pair<size_t, std::string>> chroms;
  for (auto &e : by_key) chroms.push_back(e.second);
  std::sort(chroms.begin(), chroms.end(), std::greater<std::pair<size_t, std::string>>());
  std::vector<std::vector<std::string>> groups;
  int filled = 0;  // (an int in the reference as well)
  for (auto &c : chroms) {
    if (groups.empty() || filled + c.first > (size_t)(100 * 1000 * 1000)) {
      groups.push_back({c.second});
      filled = (int)c.first;
    } else {
      groups.back().push_back(c.second);
      filled += (int)c.first;
    }
  }
  return groups;
}

static std::vector<std::string> seed_files(const std::string &path) {  // src/align_main.cc:44-62
  struct stat st;
  if (stat(path.c_str(), &st) != 0) throw "Path " + path + " is neither file nor directory";
  std::vector<std::string> files;
  if (S_ISREG(st.st_mode)) {
    files.push_back(path);
  } else if (S_ISDIR(st.st_mode)) {
    glob_t g;
    glob((path + "/*.bed").c_str(), GLOB_TILDE, nullptr, &g);
    for (size_t i = 0; i < g.gl_pathc; i++) {
      struct stat s2;
      if (stat(g.gl_pathv[i], &s2) == 0 && S_ISREG(s2.st_mode)) files.push_back(g.gl_pathv[i]);
    }
    globfree(&g);
  } else {
    throw "Path " + path + " is neither file nor directory";
  }
  return files;
}

void bucket_alignments_extern(const std::string &bed_path, int nbins, const std::string &output_dir, bool extend,
                              const std::string &reference, const BucketParams &bp, FILE *log) {
  std::map<std::string, int> group_of;
  {
    const auto groups = generate_translation(reference);
    for (int g = 0; g < (int)groups.size(); g++)
      for (auto &name : groups[(size_t)g]) group_of[name] = g;
  }
  // 1. seeds -> extended, side-ordered hits, grouped by the pair of chromosome groups.  The key is the reference's
  //    temporary file name: groups are processed in the order of these strings.
  std::map<std::string, std::vector<std::string>> group_lines;
  int total = 0;
  for (auto &file : seed_files(bed_path)) {
    std::ifstream fin(file.c_str());
    if (!fin.is_open()) throw "BED file " + bed_path + " does not exist";
    int count = 0;
    for (std::string line; std::getline(fin, line); count++) {
      Hit h = Hit::from_bed(line);
      if (extend) h.extend(bp.extend_ratio, bp.max_extend);
      order_sides(h);
      const std::string key = output_dir + "/tmp_" + std::to_string(group_of[h.query->name]) + "_" +
                              std::to_string(group_of[h.ref->name]) + ".tmp";
      group_lines[key].push_back(h.to_bed(false));
    }
    total += count;
    fprintf(log, "\rRead %10d alignments in %s         ", count, file.c_str());
  }
  fprintf(log, "\nRead total %d alignments\n", total);

  // 2. merge every group; count the hits per complexity class (sqrt of the area, in thousands)
  int max_complexity = 0;
  std::map<int, int> per_class;
  for (auto &g : group_lines) {
    std::vector<Hit> hits;
    hits.reserve(g.second.size());
    for (auto &line : g.second) hits.push_back(Hit::from_bed(line));
    if (extend) hits = merge_hits(hits, bp.merge_dist);
    g.second.clear();
    for (auto &h : hits) {
      max_complexity = std::max(max_complexity, complexity_of(h));
      per_class[complexity_class(h)]++;
      g.second.push_back(h.to_bed(false));
    }
  }
  fprintf(log, "\nFinished with sorting\n");

  // 3. deal: every complexity class goes round the buckets on its own, starting where the class below it stopped
  //    (src/align_main.cc:147-175), so that each bucket gets its share of cheap and of expensive pairs
  std::vector<int> next_bucket(1, 0);
  for (int c = 1; c <= max_complexity / 1000; c++) next_bucket.push_back((next_bucket[(size_t)c - 1] + per_class[c - 1]) % nbins);
  std::vector<FILE *> out;
  for (int b = 0; b < nbins; b++) {
    char name[64];
    snprintf(name, sizeof name, "/bucket_%04d", b);
    FILE *f = fopen((output_dir + name).c_str(), "w");
    if (!f) throw "Cannot open file " + output_dir + name + " for writing";
    out.push_back(f);
  }
  for (auto &g : group_lines)
    for (auto &line : g.second) {
      Hit h = Hit::from_bed(line);
      int &turn = next_bucket[(size_t)complexity_class(h)];
      FILE *f = out[(size_t)turn];
      turn = (turn + 1) % nbins;
      if (h.query->is_rc) {  // the reverse strand belongs on the reference side
        std::swap(h.query, h.ref);
        std::swap(h.query_start, h.ref_start);
        std::swap(h.query_end, h.ref_end);
      }
      fputs(h.to_bed(false).c_str(), f);
      fputs("\n", f);
    }
  for (auto f : out) fclose(f);
}

}  // namespace sdfh
