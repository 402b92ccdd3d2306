// tests/emu/tsan_main.cpp -- TEST INFRASTRUCTURE: runs the CPU emulation of the solve kernel (bmpc_emu.cpp) under
// ThreadSanitizer.  The emulation synchronises lanes only where the GPU does (workgroup barrier = std::barrier, pair
// exchange = a two-lane rendezvous, wave reduction = a 64-lane barrier), so an LDS hand-over between lanes that no
// s_barrier orders shows up as a data race on the shared-memory image.  Input: a dump written by tests/emu/tsan.py.
#include "bmpc_emu.cpp"

#include <cstdio>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* fd = std::fopen(argv[1], "rb");
  if (!fd) return 2;
  bmpc_params p;
  int32_t B = 0, has_mu = 0;
  if (std::fread(&p, sizeof(p), 1, fd) != 1 || std::fread(&B, 4, 1, fd) != 1 || std::fread(&has_mu, 4, 1, fd) != 1) return 2;
  const int h = p.h;
  std::vector<float> x_fb(B * 12), foot(B * 6), x_cmd(B * 12), mu(has_mu ? B * h * 2 : 0);
  std::vector<uint8_t> contact(B * h * 2);
  std::vector<int32_t> phase(B);
  bool ok = std::fread(x_fb.data(), 4, x_fb.size(), fd) == x_fb.size() && std::fread(foot.data(), 4, foot.size(), fd) == foot.size() &&
            std::fread(contact.data(), 1, contact.size(), fd) == contact.size() &&
            std::fread(phase.data(), 4, phase.size(), fd) == phase.size() && std::fread(x_cmd.data(), 4, x_cmd.size(), fd) == x_cmd.size();
  if (has_mu) ok = ok && std::fread(mu.data(), 4, mu.size(), fd) == mu.size();
  std::fclose(fd);
  if (!ok) return 2;
  std::vector<float> controls(B * h * 12), states(B * h * 13), resid(B * 2);
  std::vector<int32_t> iters(B), status(B), nfac(B);
  const int rc = bmpc_emu_solve(&p, B, x_fb.data(), foot.data(), contact.data(), phase.data(), x_cmd.data(), has_mu ? mu.data() : nullptr,
                                controls.data(), states.data(), iters.data(), resid.data(), status.data(), nfac.data(), nullptr,
                                nullptr, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0.5);
  for (int b = 0; b < B; ++b) std::printf("instance %d: iters %d status %d nfactor %d u0[2] %.6f\n", b, iters[b], status[b], nfac[b], controls[b * h * 12 + 2]);
  return rc;
}
