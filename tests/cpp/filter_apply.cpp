// Applies a DataPointsFilters YAML list of the drop-in's PointMatcher shim to a cloud read from a file and writes the result:
//   filter_apply f32|f64 FILTERS.yaml IN.bin OUT.bin
// IN.bin: int32 n, then n x 3 values (T); OUT.bin: int32 n_out, int32 has_normals, int32 has_densities, n_out x 3 points,
// [n_out x 3 normals], [n_out densities], int32 has_simpleSensorNoise, [n_out values].  The Python tests compare it with the oracle's statement of the same filters
// (tests/test_filters_host.py: host-only filters, no device; tests/test_gpu_filters.py: the ones that search neighbours on the device).
#include <pointmatcher/PointMatcher.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

template <typename T>
static int run(const char *yaml, const char *in, const char *out)
{
    using PM = PointMatcher<T>;
    std::ifstream fy(yaml);
    typename PM::DataPointsFilters filters(fy);
    FILE *fi = std::fopen(in, "rb");
    if (!fi) return 2;
    int n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 2;
    std::vector<T> xyz((size_t)3 * n);
    if (std::fread(xyz.data(), sizeof(T), xyz.size(), fi) != xyz.size()) return 2;
    std::fclose(fi);
    auto cloud = PM::DataPoints::fromXYZ(xyz.data(), n, nullptr);
    filters.init();
    filters.apply(cloud);
    const int m = (int)cloud.getNbPoints(), hn = cloud.descriptorExists("normals") ? 1 : 0, hd = cloud.descriptorExists("densities") ? 1 : 0;
    FILE *fo = std::fopen(out, "wb");
    std::fwrite(&m, 4, 1, fo); std::fwrite(&hn, 4, 1, fo); std::fwrite(&hd, 4, 1, fo);
    for (int j = 0; j < m; j++) { const T p[3] = {cloud.features(0, j), cloud.features(1, j), cloud.features(2, j)}; std::fwrite(p, sizeof(T), 3, fo); }
    if (hn) { const int r = cloud.getDescriptorStartingRow("normals"); for (int j = 0; j < m; j++) { const T p[3] = {cloud.descriptors(r, j), cloud.descriptors(r + 1, j), cloud.descriptors(r + 2, j)}; std::fwrite(p, sizeof(T), 3, fo); } }
    if (hd) { const int r = cloud.getDescriptorStartingRow("densities"); for (int j = 0; j < m; j++) { const T v = cloud.descriptors(r, j); std::fwrite(&v, sizeof(T), 1, fo); } }
    // (appended: int32 has_simpleSensorNoise, then n_out values)
    const int hs = cloud.descriptorExists("simpleSensorNoise") ? 1 : 0;
    std::fwrite(&hs, 4, 1, fo);
    if (hs) { const int r = cloud.getDescriptorStartingRow("simpleSensorNoise"); for (int j = 0; j < m; j++) { const T v = cloud.descriptors(r, j); std::fwrite(&v, sizeof(T), 1, fo); } }
    std::fclose(fo);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 5) { std::fprintf(stderr, "usage: filter_apply f32|f64 FILTERS.yaml IN.bin OUT.bin\n"); return 1; }
    try {
        return !std::strcmp(argv[1], "f64") ? run<double>(argv[2], argv[3], argv[4]) : run<float>(argv[2], argv[3], argv[4]);
    } catch (const std::exception &e) { std::fprintf(stderr, "filter_apply: %s\n", e.what()); return 3; }
}
