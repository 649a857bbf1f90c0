// One ICP through the drop-in's PointMatcher shim, configured from a YAML chain as pgslam configures its ICP objects
// (icp_.loadFromYaml + icp_(reading, reference, T_init), /root/reference/src/pgslam/LoopCloser.hpp:73, 98):
//   icp_apply f32|f64 CHAIN.yaml READING.bin REFERENCE.bin TINIT.bin OUT.bin
// clouds: int32 n, n x 3 values (T); TINIT: 16 doubles row-major; OUT: 16 doubles (T_out row-major), int32 reference points after the
// reference filters, int32 has_normals, 2 doubles: getOverlap(), getWeightedPointUsedRatio().  The reference's `normals` come from
// the chain's own referenceDataPointsFilters.
#include <pointmatcher/PointMatcher.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

template <typename T>
static typename PointMatcher<T>::DataPoints read_cloud(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    int n = 0;
    if (std::fread(&n, 4, 1, f) != 1) throw std::runtime_error("short file");
    std::vector<T> xyz((size_t)3 * n);
    if (std::fread(xyz.data(), sizeof(T), xyz.size(), f) != xyz.size()) throw std::runtime_error("short file");
    std::fclose(f);
    return PointMatcher<T>::DataPoints::fromXYZ(xyz.data(), n, nullptr);
}

template <typename T>
static int run(char **a)
{
    using PM = PointMatcher<T>;
    typename PM::ICP icp;
    if (!std::strcmp(a[2], "default")) icp.setDefault();
    else { std::ifstream fy(a[2]); icp.loadFromYaml(fy); }
    auto reading = read_cloud<T>(a[3]), reference = read_cloud<T>(a[4]);
    double T16[16];
    { FILE *f = std::fopen(a[5], "rb"); if (!f || std::fread(T16, 8, 16, f) != 16) return 2; std::fclose(f); }
    typename PM::Matrix Tin(4, 4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tin(i, j) = (T)T16[4 * i + j];
    // what the reference filters leave (the same filters once more on a copy: they are deterministic)
    auto ref_copy = reference;
    icp.referenceDataPointsFilters.init();
    icp.referenceDataPointsFilters.apply(ref_copy);
    icp.referenceDataPointsFilters.init();
    const typename PM::Matrix Tout = icp(reading, reference, Tin);
    double out[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (double)Tout(i, j);
    FILE *fo = std::fopen(a[6], "wb");
    std::fwrite(out, 8, 16, fo);
    const int m = (int)ref_copy.getNbPoints(), hn = ref_copy.descriptorExists("normals") ? 1 : 0;
    std::fwrite(&m, 4, 1, fo); std::fwrite(&hn, 4, 1, fo);
    // what pgslam reads after the ICP (Localizer.hpp:278, LoopCloser.hpp:331) and the plain ratio beside it
    const double ov[2] = {(double)icp.errorMinimizer->getOverlap(), (double)icp.errorMinimizer->getWeightedPointUsedRatio()};
    std::fwrite(ov, 8, 2, fo);
    std::fclose(fo);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 7) { std::fprintf(stderr, "usage: icp_apply f32|f64 CHAIN.yaml|default READING.bin REFERENCE.bin TINIT.bin OUT.bin\n"); return 1; }
    try { return !std::strcmp(argv[1], "f64") ? run<double>(argv) : run<float>(argv); }
    catch (const std::exception &e) { std::fprintf(stderr, "icp_apply: %s\n", e.what()); return 3; }
}
