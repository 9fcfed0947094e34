// `sedef align bucket` (restates reference src/align_main.cc:38-198, src/merge.cc:35-109,
// src/search_main.cc:93-120): extend the seed hits, canonicalise, spill per chromosome-group pair, merge
// nearby hits, and deal them round-robin per complexity class into N bucket files.
#include <dirent.h>
#include <glob.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <tuple>

#include "sedef_host.h"

namespace sdfh {

std::vector<Hit> merge_hits(std::vector<Hit> &hits, int merge_dist) {  // src/merge.cc:35-109
  std::vector<Hit> results;
  for (auto &h : hits) {
    if (std::tie(h.query->name, h.query_start, h.query_end) > std::tie(h.ref->name, h.ref_start, h.ref_end)) {
      std::swap(h.query->name, h.ref->name);
      std::swap(h.query_start, h.ref_start);
      std::swap(h.query_end, h.ref_end);
    }
  }
  // same library sort, same comparator, same input order => same permutation among equal keys
  std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) {
    return std::tie(a.ref->is_rc, a.query->name, a.ref->name, a.query_start, a.ref_start) <
           std::tie(b.ref->is_rc, b.query->name, b.ref->name, b.query_start, b.ref_start);
  });
  Hit prev;
  std::multimap<int, Hit> windows;
  for (auto &rec : hits) {
    if (rec.query->name == rec.ref->name && rec.query_start == rec.ref_start && rec.query_end == rec.ref_end &&
        rec.query->is_rc == rec.ref->is_rc)
      continue;
    if ((&rec - &hits[0]) == 0) {
      windows.emplace(rec.ref_end, rec);
      prev = rec;
    } else if (prev.query_end + merge_dist < rec.query_start || prev.query->name != rec.query->name ||
               prev.ref->name != rec.ref->name || prev.ref->is_rc != rec.ref->is_rc) {
      for (auto &it : windows) results.push_back(it.second);
      windows.clear();
      windows.emplace(rec.ref_end, rec);
      prev = rec;
    } else {
      bool need_update = true;
      while (need_update) {
        auto loc = windows.lower_bound(rec.ref_start - merge_dist);
        need_update = false;
        while (loc != windows.end()) {
          if (loc->second.query_end + merge_dist < rec.query_start || loc->second.ref_end < rec.ref_start - merge_dist ||
              loc->second.ref_start > rec.ref_end + merge_dist) {
            ++loc;
            continue;
          }
          need_update = true;
          rec.query_end = std::max(rec.query_end, loc->second.query_end);
          rec.ref_end = std::max(rec.ref_end, loc->second.ref_end);
          rec.query_start = std::min(rec.query_start, loc->second.query_start);
          rec.ref_start = std::min(rec.ref_start, loc->second.ref_start);
          windows.erase(loc++);
        }
      }
      windows.emplace(rec.ref_end, rec);
    }
    rec.query_end = std::max(rec.query_end, prev.query_end);
    prev = rec;
  }
  for (auto &it : windows) results.push_back(it.second);
  return results;
}

// chromosomes sorted by (length, name) descending, grouped greedily into <= 100 MB groups
std::vector<std::vector<std::string>> generate_translation(const std::string &ref_path) {  // src/search_main.cc:93-120
  std::ifstream fin((ref_path + ".fai").c_str());
  if (!fin.is_open()) throw "Index file " + ref_path + ".fai does not exist";
  std::map<std::string, std::pair<size_t, std::string>> index;  // keyed by first token, like FastaIndex
  std::string line;
  while (std::getline(fin, line)) {
    auto f = split(line, '\t');
    if (f.size() != 5) throw "Index file " + ref_path + ".fai is malformed";
    index.insert({split(f[0], ' ').at(0), {(size_t)atoi(f[1].c_str()), f[0]}});
  }
  std::vector<std::pair<size_t, std::string>> vv;
  for (auto &e : index) vv.push_back(e.second);
  std::sort(vv.begin(), vv.end(), std::greater<std::pair<size_t, std::string>>());
  std::vector<std::vector<std::string>> ref;
  int cur_size = 0;
  const int MAX_SIZE = 100 * 1000 * 1000;
  for (auto &v : vv) {
    if (ref.empty() || cur_size + v.first > (size_t)MAX_SIZE) {
      ref.push_back({v.second});
      cur_size = (int)v.first;
    } else {
      ref.back().push_back(v.second);
      cur_size += (int)v.first;
    }
  }
  return ref;
}

static std::vector<std::string> list_beds(const std::string &bed_path) {
  struct stat st;
  if (stat(bed_path.c_str(), &st) != 0) throw "Path " + bed_path + " is neither file nor directory";
  std::vector<std::string> files;
  if (S_ISREG(st.st_mode)) {
    files.push_back(bed_path);
  } else if (S_ISDIR(st.st_mode)) {
    glob_t g;
    glob((bed_path + "/*.bed").c_str(), GLOB_TILDE, nullptr, &g);
    for (size_t i = 0; i < g.gl_pathc; i++) {
      struct stat s2;
      if (stat(g.gl_pathv[i], &s2) == 0 && S_ISREG(s2.st_mode)) files.push_back(g.gl_pathv[i]);
    }
    globfree(&g);
  } else {
    throw "Path " + bed_path + " is neither file nor directory";
  }
  return files;
}

void bucket_alignments_extern(const std::string &bed_path, int nbins, const std::string &output_dir, bool extend,
                              const std::string &reference, const BucketParams &bp, FILE *log) {
  auto ref = generate_translation(reference);
  std::map<std::string, int> lookup;
  for (int i = 0; i < (int)ref.size(); i++)
    for (auto &j : ref[i]) lookup[j] = i;

  std::map<std::string, FILE *> tmp_bins;
  std::map<std::string, int> lens;
  int total_nhits = 0;
  for (auto &file : list_beds(bed_path)) {
    std::ifstream fin(file.c_str());
    if (!fin.is_open()) throw "BED file " + bed_path + " does not exist";
    std::string s;
    int nhits = 0;
    while (std::getline(fin, s)) {
      Hit h = Hit::from_bed(s);
      if (extend) h.extend(bp.extend_ratio, bp.max_extend);
      if (std::tie(h.query->name, h.query_start, h.query_end) > std::tie(h.ref->name, h.ref_start, h.ref_end)) {
        std::swap(h.query->name, h.ref->name);
        std::swap(h.query_start, h.ref_start);
        std::swap(h.query_end, h.ref_end);
      }
      const std::string fno = output_dir + "/tmp_" + std::to_string(lookup[h.query->name]) + "_" +
                              std::to_string(lookup[h.ref->name]) + ".tmp";
      auto it = tmp_bins.find(fno);
      if (it == tmp_bins.end()) {
        FILE *f = fopen(fno.c_str(), "w");
        if (!f) throw "Cannot open file " + fno + " for writing";
        it = tmp_bins.emplace(fno, f).first;
      }
      fputs(h.to_bed(false).c_str(), it->second);
      fputs("\n", it->second);
      lens[fno]++;
      nhits++;
      total_nhits++;
    }
    fprintf(log, "\rRead %10d alignments in %s         ", nhits, file.c_str());
  }
  fprintf(log, "\nRead total %d alignments\n", total_nhits);

  int max_complexity = 0;
  std::map<int, int> complexity;
  for (auto &bin : tmp_bins) {
    fclose(bin.second);
    std::ifstream fin(bin.first.c_str());
    std::vector<Hit> hits;
    hits.reserve(lens[bin.first]);
    std::string s;
    while (std::getline(fin, s)) hits.push_back(Hit::from_bed(s));
    fin.close();
    if (extend) hits = merge_hits(hits, bp.merge_dist);
    for (auto &h : hits) {
      const int c = (int)std::sqrt(double(h.query_end - h.query_start) * double(h.ref_end - h.ref_start));
      max_complexity = std::max(max_complexity, c);
      complexity[c / 1000]++;
    }
    FILE *fo = fopen(bin.first.c_str(), "w");
    for (auto &h : hits) {
      fputs(h.to_bed(false).c_str(), fo);
      fputs("\n", fo);
    }
    fclose(fo);
  }
  fprintf(log, "\nFinished with sorting\n");

  std::vector<int> next_bin(1, 0);
  for (int c = 1; c <= max_complexity / 1000; c++) next_bin.push_back((next_bin[c - 1] + complexity[c - 1]) % nbins);

  std::vector<FILE *> fout;
  for (int b = 0; b < nbins; b++) {
    char name[64];
    snprintf(name, sizeof name, "/bucket_%04d", b);
    FILE *f = fopen((output_dir + name).c_str(), "w");
    if (!f) throw "Cannot open file " + output_dir + name + " for writing";
    fout.push_back(f);
  }
  for (auto &bin : tmp_bins) {
    std::ifstream fin(bin.first.c_str());
    std::string s;
    while (std::getline(fin, s)) {
      Hit h = Hit::from_bed(s);
      int cx = (int)std::sqrt(double(h.query_end - h.query_start) * double(h.ref_end - h.ref_start));
      cx /= 1000;
      const int b = next_bin[cx];
      next_bin[cx] = (next_bin[cx] + 1) % nbins;
      if (h.query->is_rc) {
        std::swap(h.query, h.ref);
        std::swap(h.query_start, h.ref_start);
        std::swap(h.query_end, h.ref_end);
      }
      fputs(h.to_bed(false).c_str(), fout[b]);  // buffered in blocks of 1000 in the reference: same file content
      fputs("\n", fout[b]);
    }
  }
  for (auto f : fout) fclose(f);
  for (auto &s : tmp_bins) unlink(s.first.c_str());
}

}  // namespace sdfh
