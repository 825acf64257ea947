// pwrite_files.cpp -- does writing a plotfile level into ONE file serialise the writer's threads?  T threads pwrite disjoint 32 MiB chunks
// (total G GiB) into 1 file or into F files.   build: g++ -O2 -pthread pwrite_files.cpp -o pwrite_files    usage: pwrite_files dir GiB threads
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const long long G = argc > 2 ? atoll(argv[2]) : 8;
  const int T = argc > 3 ? atoi(argv[3]) : 16;
  const size_t chunk = 32u << 20;
  const long long nchunks = G * (1LL << 30) / (long long)chunk;
  std::vector<char> buf(chunk);
  for (size_t i = 0; i < chunk; ++i) buf[i] = (char)(i * 7);
  for (int F : {1, 4, 16}) {
    std::vector<int> fd(F);
    for (int f = 0; f < F; ++f) {
      const std::string p = dir + "/pwf_" + std::to_string(F) + "_" + std::to_string(f);
      fd[f] = open(p.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
      if (fd[f] < 0) { perror("open"); return 1; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        for (long long c = t; c < nchunks; c += T) {
          const int f = (int)(c % F);
          const long long off = (c / F) * (long long)chunk;
          size_t done = 0;
          while (done < chunk) {
            const ssize_t r = pwrite(fd[f], buf.data() + done, chunk - done, (off_t)(off + (long long)done));
            if (r <= 0) { perror("pwrite"); return; }
            done += (size_t)r;
          }
        }
      });
    for (auto& x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%2d file(s), %d threads: %lld GiB in %.2f s = %.2f GiB/s\n", F, T, G, s, G / s);
    for (int f = 0; f < F; ++f) {
      close(fd[f]);
      unlink((dir + "/pwf_" + std::to_string(F) + "_" + std::to_string(f)).c_str());
    }
  }
  return 0;
}
