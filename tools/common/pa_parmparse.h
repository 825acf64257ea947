// pa_parmparse.h -- the slice of amrex::ParmParse the PeleAnalysis tools use:
//   exe [inputs_file] key=value ...    values are whitespace separated lists; '#' starts a comment
//   in files; when a name is defined more than once the LAST definition wins; get() aborts if the
//   name is absent, query() leaves the variable untouched.  amrex.* / fab.* keys are ignored.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace pa {

[[noreturn]] inline void Abort(const std::string& msg) {
  // same text as amrex::Abort
  std::cerr << "amrex::Abort::0::" << msg << " !!!" << std::endl;
  std::fflush(nullptr);
  // _Exit, not exit: Abort is also reached from plotfile worker threads while other workers (and the thread that
  // brings up the HIP runtime) are still running; exit() would run static destructors under live threads.
  // amrex::Abort itself ends in std::abort() -- no destructors either.
  std::_Exit(134);
}

// Normal end of a tool: everything it writes has been written and closed.  Leaves without unwinding: tearing down the HIP
// runtime, the device allocations and GBs of host buffers costs 0.1-0.3 s that produce nothing (the OS reclaims all of it).
[[noreturn]] inline void Finish() {
  std::cout.flush();
  std::cerr.flush();
  std::fflush(nullptr);
  if (const char* e = std::getenv("PA_TOOL_EXIT")) {  // "normal": run the exit handlers (a profiler attached to the tool flushes its trace there)
    if (std::string(e) == "normal") std::exit(0);
  }
  std::_Exit(0);
}

class ParmParse {
 public:
  ParmParse(int argc, char** argv) {
    int first = 1;
    if (argc > 1 && std::string(argv[1]).find('=') == std::string::npos) {
      std::ifstream f(argv[1]);
      if (!f) Abort(std::string("ParmParse: cannot open inputs file ") + argv[1]);
      std::stringstream ss;
      std::string line;
      while (std::getline(f, line)) {
        const auto h = line.find('#');
        if (h != std::string::npos) line.erase(h);
        ss << line << '\n';
      }
      parse(ss.str());
      first = 2;
    }
    std::string all;
    for (int i = first; i < argc; ++i) { all += argv[i]; all += '\n'; }
    parse(all);
  }
  bool contains(const std::string& n) const { return tab_.count(n) > 0; }
  int countval(const std::string& n) const { auto it = tab_.find(n); return it == tab_.end() ? 0 : (int)it->second.size(); }

  template <typename T> void get(const std::string& n, T& v, int idx = 0) const {
    auto it = tab_.find(n);
    if (it == tab_.end() || idx >= (int)it->second.size()) Abort("ParmParse::get(): " + n + " not found in table");
    conv(n, it->second[idx], v);
  }
  template <typename T> bool query(const std::string& n, T& v, int idx = 0) const {
    auto it = tab_.find(n);
    if (it == tab_.end() || idx >= (int)it->second.size()) return false;
    conv(n, it->second[idx], v);
    return true;
  }
  template <typename T> bool queryarr(const std::string& n, std::vector<T>& v, int start, int num) const {
    auto it = tab_.find(n);
    if (it == tab_.end()) return false;
    if (start + num > (int)it->second.size()) Abort("ParmParse::queryarr(): " + n + " has too few values");
    v.resize(num);
    for (int i = 0; i < num; ++i) conv(n, it->second[start + i], v[i]);
    return true;
  }
  template <typename T> void getarr(const std::string& n, std::vector<T>& v) const {
    auto it = tab_.find(n);
    if (it == tab_.end()) Abort("ParmParse::getarr(): " + n + " not found in table");
    v.resize(it->second.size());
    for (size_t i = 0; i < v.size(); ++i) conv(n, it->second[i], v[i]);
  }

 private:
  std::map<std::string, std::vector<std::string>> tab_;

  void parse(const std::string& text) {
    // tokens: name = v1 v2 ... ; a new definition starts at a token followed by '='
    std::vector<std::string> tok;
    std::string cur;
    bool quoted = false;
    auto flush = [&]() { if (!cur.empty()) { tok.push_back(cur); cur.clear(); } };
    for (char c : text) {
      if (c == '"') { quoted = !quoted; continue; }
      if (!quoted && (c == ' ' || c == '\t' || c == '\n' || c == '\r')) { flush(); continue; }
      if (!quoted && c == '=') { flush(); tok.push_back("="); continue; }
      cur += c;
    }
    flush();
    for (size_t i = 0; i < tok.size();) {
      if (i + 1 < tok.size() && tok[i + 1] == "=") {
        const std::string name = tok[i];
        i += 2;
        std::vector<std::string> vals;
        while (i < tok.size() && !(i + 1 < tok.size() && tok[i + 1] == "=")) vals.push_back(tok[i++]);
        if (name.rfind("amrex.", 0) == 0 || name.rfind("fab.", 0) == 0) continue;
        tab_[name] = vals;  // last definition wins
      } else {
        ++i;  // stray token
      }
    }
  }
  static void conv(const std::string& n, const std::string& s, std::string& v) { (void)n; v = s; }
  static void conv(const std::string& n, const std::string& s, int& v) {
    char* e = nullptr;
    v = (int)std::strtol(s.c_str(), &e, 10);
    if (e == s.c_str() || *e) {
      if (s == "true" || s == "t") v = 1;
      else if (s == "false" || s == "f") v = 0;
      else Abort("ParmParse: " + n + ": expected an int, got " + s);
    }
  }
  static void conv(const std::string& n, const std::string& s, bool& v) {
    if (s == "true" || s == "t" || s == "1") v = true;
    else if (s == "false" || s == "f" || s == "0") v = false;
    else Abort("ParmParse: " + n + ": expected a bool, got " + s);
  }
  static void conv(const std::string& n, const std::string& s, double& v) {
    char* e = nullptr;
    v = std::strtod(s.c_str(), &e);
    if (e == s.c_str() || *e) Abort("ParmParse: " + n + ": expected a real, got " + s);
  }
};

inline std::string getFileRoot(const std::string& infile) {  // grad.cpp:26-31
  std::string s = infile;
  while (!s.empty() && s.back() == '/') s.pop_back();
  const auto p = s.rfind('/');
  return p == std::string::npos ? s : s.substr(p + 1);
}

}  // namespace pa
