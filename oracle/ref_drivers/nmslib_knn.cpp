// nmslib_knn.cpp -- driver for the reference's vendored NMSLIB (the engine of its BRUTEFORCENMS matcher,
// matchinglib/source/matchers.cpp:476-519 via matchinglib/include/nmslib/nmslib_matchers.h:159-424).
// TEST INFRASTRUCTURE: built only in the container that has /root/reference, by oracle/Makefile, against the
// NMSLIB sources where they lie; used to pin the CPU restatement (oracle/knn_oracle.c) and to generate the
// golden vectors under tests/golden/.  This file is ours; it contains no reference source.
//
// usage: nmslib_knn <hamming|hamming_wrapper|l2> <in.bin> <out.bin>
//   hamming_wrapper packs the descriptors exactly like the reference's wrapper (nmslib_matchers.h:214-231): two bytes per
//   int (b[j] << 8 | b[j+1], an odd last byte alone), Object created WITHOUT the trailing length word, so that
//   SpaceBitHamming::HiddenDistance (length = datalength/4 - 1) drops the last int -- the 240-bit quirk of BRUTEFORCENMS.
//   in.bin : int32 nq, nt, width ; then nq*width and nt*width elements (uint8 for hamming, float32 for l2)
//   out.bin: int32 idx[nq][2] ; then dist[nq][2] (int32 for hamming; float32 = NMSLIB's sqrt L2 for l2)
// Unlike the matchinglib wrapper (which packs 2 bytes per int and omits the trailing length word, so the last
// bytes are ignored), this driver hands NMSLIB the full bit string: 32-bit words + the length word that
// SpaceBitHamming::HiddenDistance strips (similarity_search/src/space/space_bit_hamming.cc:33-42).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "init.h"
#include "index.h"
#include "knnquery.h"
#include "knnqueue.h"
#include "methodfactory.h"
#include "object.h"
#include "params.h"
#include "space.h"
#include "space/space_bit_hamming.h"
#include "space/space_vector.h"
#include "spacefactory.h"

using namespace similarity;

template <typename T>
static std::vector<T> read_all(FILE *f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return v;
}

template <typename dist_t>
static void run_queries(Space<dist_t> &space, Index<dist_t> &index, const ObjectVector &queries, std::vector<int32_t> &idx,
                        std::vector<dist_t> &dist) {
    const unsigned K = 2;
    for (size_t q = 0; q < queries.size(); ++q) {
        KNNQuery<dist_t> knn(space, queries[q], K);
        index.Search(&knn);
        std::unique_ptr<KNNQueue<dist_t>> res(knn.Result()->Clone());
        // queue pops farthest first
        int pos = (int)res->Size() - 1;
        while (!res->Empty()) {
            idx[q * K + pos] = res->TopObject()->id();
            dist[q * K + pos] = res->TopDistance();
            res->Pop();
            --pos;
        }
    }
}

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s <hamming|l2> in.bin out.bin\n", argv[0]);
        return 1;
    }
    const std::string mode = argv[1];
    const bool wrapper = mode == "hamming_wrapper";
    const bool hamming = mode == "hamming" || wrapper;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 2;
    const int nq = hdr[0], nt = hdr[1], width = hdr[2];
    initLibrary(LIB_LOGNONE, nullptr);
    AnyParams empty;
    std::vector<int32_t> idx((size_t)nq * 2, -1);
    FILE *out = nullptr;
    if (hamming) {
        std::vector<uint8_t> q = read_all<uint8_t>(f, (size_t)nq * width);
        std::vector<uint8_t> t = read_all<uint8_t>(f, (size_t)nt * width);
        std::unique_ptr<Space<int>> space(SpaceFactoryRegistry<int>::Instance().CreateSpace("bit_hamming", empty));
        SpaceBitHamming *bh = dynamic_cast<SpaceBitHamming *>(space.get());
        const int nw = (width + 3) / 4;
        auto make = [&](const uint8_t *row, int id) -> Object * {
            if (wrapper) {
                std::vector<int> v;
                for (int j = 0; j < width - 1; j += 2) v.push_back(((int)row[j] << 8) | (int)row[j + 1]);
                if (width % 2) v.push_back((int)row[width - 1]);
                return new Object(id, 0, v.size() * sizeof(int), &v[0]);
            }
            std::vector<uint32_t> w(nw, 0u);
            memcpy(w.data(), row, width);
            return bh->CreateObjFromVect(id, 0, w);  // appends the length word
        };
        ObjectVector data, queries;
        for (int i = 0; i < nt; ++i) data.push_back(make(&t[(size_t)i * width], i));
        for (int i = 0; i < nq; ++i) queries.push_back(make(&q[(size_t)i * width], i));
        std::unique_ptr<Index<int>> index(
            MethodFactoryRegistry<int>::Instance().CreateMethod(false, "seq_search", "bit_hamming", *space, data));
        index->CreateIndex(empty);
        std::vector<int> dist((size_t)nq * 2, -1);
        run_queries<int>(*space, *index, queries, idx, dist);
        out = fopen(argv[3], "wb");
        fwrite(idx.data(), 4, idx.size(), out);
        fwrite(dist.data(), 4, dist.size(), out);
    } else {
        std::vector<float> q = read_all<float>(f, (size_t)nq * width);
        std::vector<float> t = read_all<float>(f, (size_t)nt * width);
        std::unique_ptr<Space<float>> space(SpaceFactoryRegistry<float>::Instance().CreateSpace("l2", empty));
        VectorSpace<float> *vs = dynamic_cast<VectorSpace<float> *>(space.get());
        auto make = [&](const float *row, int id) {
            std::vector<float> v(row, row + width);
            return vs->CreateObjFromVect(id, 0, v);
        };
        ObjectVector data, queries;
        for (int i = 0; i < nt; ++i) data.push_back(make(&t[(size_t)i * width], i));
        for (int i = 0; i < nq; ++i) queries.push_back(make(&q[(size_t)i * width], i));
        std::unique_ptr<Index<float>> index(
            MethodFactoryRegistry<float>::Instance().CreateMethod(false, "seq_search", "l2", *space, data));
        index->CreateIndex(empty);
        std::vector<float> dist((size_t)nq * 2, -1.f);
        run_queries<float>(*space, *index, queries, idx, dist);
        out = fopen(argv[3], "wb");
        fwrite(idx.data(), 4, idx.size(), out);
        fwrite(dist.data(), 4, dist.size(), out);
    }
    fclose(out);
    fclose(f);
    return 0;
}
