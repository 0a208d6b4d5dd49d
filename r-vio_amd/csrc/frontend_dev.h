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
    float* cell_pts;        // ChessGrid cells           [cells][2F][2]
    // outputs: mvFeatTypesForUpdate / mvlFeatMeasForUpdate
    int* n_feat;
    unsigned char* types;   // [Fu]
    int* len;               // [Fu]
    float* meas;            // [Fu][max_len][2]
    rvio_frame_info* info;
};
