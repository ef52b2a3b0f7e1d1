// C++ client of a deployed TorchScript archive -- what LAMMPS' pair style does with a `spkdeploy`ed model
// (interfaces/lammps/pair_schnetpack.cpp:125-131: torch::jit::load(path, device, metadata) + metadata["cutoff"];
// :285-328: the input dict of one system and model.forward), plus the one thing this package adds to the build glue
// (interfaces/lammps/patch_lammps.sh:56-69): the two shared libraries that hold the spk_hip:: operators are dlopen'ed
// before the archive is loaded -- libspk_hip.so (the C ABI, RTLD_GLOBAL) and libspk_torch.so (TORCH_LIBRARY(spk_hip)).
// No Python anywhere in this process.
//
//   spk_jit_client <model.pt> <system.bin> <libspk_hip.so> <libspk_torch.so> [device]
//
// system.bin (little endian): int64 n_atoms, int64 n_edges, int64 Z[n], float R[n][3], int64 idx_i[E], int64 idx_j[E],
// float offsets[E][3], float cell[9].  Output (text): cutoff, energy, then the forces, one atom per line.
#include <dlfcn.h>
#include <torch/script.h>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

template <class T>
static std::vector<T> rd(std::ifstream& f, size_t n) {
  std::vector<T> v(n);
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
  if (!f) throw std::runtime_error("system file too short");
  return v;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s model.pt system.bin libspk_hip.so libspk_torch.so [device]\n", argv[0]);
    return 2;
  }
  try {
    if (!dlopen(argv[3], RTLD_NOW | RTLD_GLOBAL)) throw std::runtime_error(std::string("dlopen: ") + dlerror());
    if (!dlopen(argv[4], RTLD_NOW | RTLD_GLOBAL)) throw std::runtime_error(std::string("dlopen: ") + dlerror());
    const torch::Device device(argc > 5 ? argv[5] : "cuda:0");
    std::unordered_map<std::string, std::string> metadata = {{"cutoff", ""}};
    torch::jit::Module model = torch::jit::load(std::string(argv[1]), device, metadata);
    model.eval();
    const double cutoff = std::stod(metadata["cutoff"]);

    std::ifstream f(argv[2], std::ios::binary);
    if (!f) throw std::runtime_error("cannot open the system file");
    const auto hdr = rd<int64_t>(f, 2);
    const int64_t n = hdr[0], E = hdr[1];
    auto Z = rd<int64_t>(f, n);
    auto R = rd<float>(f, 3 * n);
    auto ii = rd<int64_t>(f, E);
    auto jj = rd<int64_t>(f, E);
    auto off = rd<float>(f, 3 * E);
    auto cell = rd<float>(f, 9);
    const auto i64 = torch::TensorOptions().dtype(torch::kInt64);
    c10::Dict<std::string, torch::Tensor> input;
    input.insert("_positions", torch::from_blob(R.data(), {n, 3}).clone().to(device));
    input.insert("_idx_i", torch::from_blob(ii.data(), {E}, i64).clone().to(device));
    input.insert("_idx_j", torch::from_blob(jj.data(), {E}, i64).clone().to(device));
    input.insert("_idx_m", torch::zeros({n}, i64).to(device));
    input.insert("_offsets", torch::from_blob(off.data(), {E, 3}).clone().to(device));
    input.insert("_cell", torch::from_blob(cell.data(), {1, 3, 3}).clone().to(device));
    input.insert("_n_atoms", torch::full({1}, n, i64).to(device));
    input.insert("_atomic_numbers", torch::from_blob(Z.data(), {n}, i64).clone().to(device));
    std::vector<torch::IValue> input_vector(1, input);
    auto output = model.forward(input_vector).toGenericDict();
    torch::Tensor forces = output.at("forces").toTensor().cpu().to(torch::kFloat64).contiguous();
    torch::Tensor energy = output.at("energy").toTensor().cpu().to(torch::kFloat64).reshape({-1});
    std::printf("cutoff %.9g\nenergy %.9f\n", cutoff, energy.data_ptr<double>()[0]);
    const double* F = forces.data_ptr<double>();
    for (int64_t a = 0; a < n; ++a) std::printf("%.9g %.9g %.9g\n", F[3 * a], F[3 * a + 1], F[3 * a + 2]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "spk_jit_client: %s\n", e.what());
    return 1;
  }
  return 0;
}
