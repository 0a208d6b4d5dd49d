// frontend_dev.h — device-side views of the tracker state (all pointers into HBM
// owned by the rvio_hip handle).
#pragma once
#include <stdint.h>

struct rvio_frame_info;

// one image pyramid: u8 levels (compact, stride = width) + int16 (dx,dy) Scharr derivatives
struct PyrDev {
    const uint8_t* img[4];
    const short* dxy[4];
    int w[4], h[4];
};

// Tracker members (Tracker.h:67-120) as flat device arrays
struct TrackerDev {
    int* first;             // mbIsTheFirstImage
    int* n_pts;             // mnFeatsToTrack
    float* feats;           // mvFeatsToTrack            [F][2] px
    float* un1;             // mPoints1ForRansac (x,y; z=1) [F][2]
    int* slot;              // mvInlierIndices           [F]
    float* hist;            // mvlTrackingHistory        [F][max_len][2]
    int* hist_len;          // list sizes                [F]
    // per-frame scratch
    float* tracked;         // vFeatsTracked             [F][2]
    float* un2;             // vFeatsUndistNorm          [F][2]
    unsigned char* status;  // vInlierFlag               [F]
    float* tmp_feats; float* tmp_un; int* tmp_slot;      // next-frame order being built
    int* cand_acc;          // FindNewer accept flags    [F]
    int* mid;               // book-keeping, hand-over half -> refill half: {first image?, survivors nIn, nMeas}
    float* cell_pts;        // ChessGrid cells           [cells][2F][2]
    // outputs: mvFeatTypesForUpdate / mvlFeatMeasForUpdate
    int* n_feat;
    unsigned char* types;   // [Fu]
    int* len;               // [Fu]
    float* meas;            // [Fu][max_len][2]
    rvio_frame_info* info;
    int* first_host;        // host-mapped mirror of `first`, one int per instance (NOT a slab member: indexed by blockIdx.z); the host only
                            // reads it to drop a cross-stream wait once the first image is behind every instance
};

// Batched launches (gridDim.z = instances, rvio_dev.h): every member of these views lies in the instance's slab
#ifdef __HIPCC__
template <typename T>
__device__ __forceinline__ void zmove(T*& p, size_t off) { if (p) p = (T*)((char*)p + off); }
__device__ __forceinline__ void pyr_shift(PyrDev& p, size_t off) {
#pragma unroll
    for (int l = 0; l < 4; ++l) { zmove(p.img[l], off); zmove(p.dxy[l], off); }
}
__device__ __forceinline__ void tracker_shift(TrackerDev& t, size_t off) {
    zmove(t.first, off); zmove(t.n_pts, off); zmove(t.feats, off); zmove(t.un1, off); zmove(t.slot, off); zmove(t.hist, off); zmove(t.hist_len, off);
    zmove(t.tracked, off); zmove(t.un2, off); zmove(t.status, off); zmove(t.tmp_feats, off); zmove(t.tmp_un, off); zmove(t.tmp_slot, off);
    zmove(t.cand_acc, off); zmove(t.mid, off); zmove(t.cell_pts, off); zmove(t.n_feat, off); zmove(t.types, off); zmove(t.len, off); zmove(t.meas, off); zmove(t.info, off);
}
#endif
