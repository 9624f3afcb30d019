// Host-side check of the C ABI (include/gtgvae.h) for a build whose HOST code is compiled with AddressSanitizer
// and UndefinedBehaviorSanitizer (make -C gtg-vae_amd/csrc hostcheck).  It needs no GPU: it calls what runs on
// the host alone -- the version / error / counter calls, every *_workspace_bytes function over a grid of sizes up
// to the largest the entry points accept (signed overflow in a size computation is what UBSan would report), and
// every compute entry point with arguments that its validation must refuse before anything is launched.
// Prints "host check ok: <calls> calls" and exits 0; a sanitizer report or an accepted bad call makes it fail.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/gtgvae.h"

static int g_calls = 0, g_bad = 0;

#define REFUSED(expr)                                                              \
    do {                                                                           \
        ++g_calls;                                                                 \
        const int rc_ = (expr);                                                    \
        if (rc_ >= 0) {                                                            \
            std::fprintf(stderr, "accepted: %s -> %d\n", #expr, rc_);              \
            ++g_bad;                                                               \
        } else if (std::strlen(gtg_last_error()) == 0) {                           \
            std::fprintf(stderr, "no message: %s -> %d\n", #expr, rc_);            \
            ++g_bad;                                                               \
        }                                                                          \
    } while (0)

#define SIZE(expr)                                                                 \
    do {                                                                           \
        ++g_calls;                                                                 \
        volatile size_t s_ = (expr);                                               \
        (void)s_;                                                                  \
    } while (0)

int main()
{
    if (gtg_version() != GTG_VERSION || gtg_max_nodes() != 64) return 2;
    (void)gtg_launch_count();
    char small[8];
    (void)gtg_profile_enable(0);
    (void)gtg_profile_report(small, sizeof(small));
    (void)gtg_profile_report(nullptr, 0);

    const int64_t graphs[] = {1, 2, 63, 512, 4096, GTG_MAX_GRAPHS_PER_CALL};
    const int nodes[] = {1, 2, 10, 33, 64};
    const int widths[] = {1, 3, 32, 256, 512, 1000};
    for (int64_t b : graphs)
        for (int n : nodes)
            for (int d : widths) {
                SIZE(gtg_encoder_fwd_workspace_bytes(b, n, n, d, 0));
                SIZE(gtg_encoder_fwd_workspace_bytes(b, n, n, d, 1));
                SIZE(gtg_encoder_bwd_workspace_bytes(b, n, n, d, 0));
                SIZE(gtg_encoder_bwd_workspace_bytes(b, n, n, d, 1));
                SIZE(gtg_encoder_compact_fwd_workspace_bytes(b, n, d));
                SIZE(gtg_encoder_compact_bwd_workspace_bytes(b, n, d));
                SIZE(gtg_latent_workspace_bytes(b, d));
                SIZE(gtg_decoder_bwd_workspace_bytes(b, d, 512, n * (n + 1) / 2));
                SIZE(gtg_bce_workspace_bytes(b));
            }
    SIZE(gtg_gemm_splitk_workspace_bytes(512, 1024, 32768));
    SIZE(gtg_gemm_splitk_workspace_bytes(32768, 512, 1024));
    SIZE(gtg_gemm_splitk_workspace_bytes(1, 1, 1));
    SIZE(gtg_gemm_rows_workspace_bytes());
    SIZE(gtg_proj_ce_fwd_workspace_bytes(10240, 1024, 30522));
    SIZE(gtg_proj_ce_bwd_workspace_bytes(10240, 1024, 30522));
    SIZE(gtg_proj_ce_fwd_workspace_bytes(1, 4, 2));
    SIZE(gtg_lstm_fwd_workspace_bytes(40, 256, 512, 1024));
    SIZE(gtg_lstm_bwd_workspace_bytes(40, 256, 512, 1024));
    SIZE(gtg_lstm_bwd_workspace_bytes(1, 1, 4, 64));
    SIZE(gtg_embed_dropout_bwd_workspace_bytes(10240, 30522, 512, 256));
    SIZE(gtg_embed_dropout_bwd_workspace_bytes(262144, 1 << 20, 1024, 0));
    SIZE(gtg_bce_loss_fwd_noalloc_max_graphs());

    // ---- arguments every entry point must refuse on the host -------------------------------------------------
    float f = 0.f;
    int32_t i32 = 0;
    int64_t i64 = 0;
    uint64_t u64 = 0;
    uint8_t u8 = 0;
    void* st = nullptr;
    REFUSED(gtg_stream_create(0, nullptr));
    REFUSED(gtg_stream_create(-1, &st));
    REFUSED(gtg_mst_batched_triu(nullptr, 4, 65, nullptr, nullptr, nullptr));
    REFUSED(gtg_mst_batched_triu(nullptr, 4, 10, nullptr, nullptr, nullptr));
    REFUSED(gtg_mst_batched_full(&f, -1, 10, &i32, &i32, nullptr));
    REFUSED(gtg_edge_prf(nullptr, nullptr, nullptr, 4, 10, nullptr, nullptr, nullptr));
    REFUSED(gtg_edge_prf(&i32, &i32, &u64, 4, 65, &f, &f, nullptr));
    REFUSED(gtg_adj_build(nullptr, 0, 2, 0, nullptr, nullptr, nullptr, nullptr, nullptr));
    REFUSED(gtg_adj_build(nullptr, 8, 2, 10, &i32, &i32, &f, &i32, nullptr));
    REFUSED(gtg_encoder_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 10, 10, 8, nullptr, nullptr, nullptr,
                            GTG_MATRIX_F32, nullptr, 0, nullptr));
    REFUSED(gtg_encoder_fwd(&f, nullptr, &f, &f, &f, &f, 4, 10, 10, 8, &f, &f, &i32, 99, &f, 1 << 20, nullptr));
    REFUSED(gtg_encoder_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 10, 10, 8, nullptr, nullptr,
                            nullptr, nullptr, GTG_MATRIX_F32, nullptr, 0, nullptr));
    REFUSED(gtg_encoder_compact_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 10, 8, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, GTG_MATRIX_F16X2, nullptr, 0, nullptr));
    REFUSED(gtg_encoder_compact_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 10, 8, nullptr, nullptr,
                                    nullptr, nullptr, GTG_MATRIX_F16X2, nullptr, 0, nullptr));
    REFUSED(gtg_latent_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 8, nullptr, nullptr, nullptr, nullptr,
                           nullptr, 0, nullptr));
    REFUSED(gtg_latent_fwd(&f, &f, &f, &f, &f, &f, 0, 8, &f, &f, &f, &f, &f, 4, nullptr));
    REFUSED(gtg_latent_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 4, 8, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_reparam_kl_fwd(nullptr, nullptr, nullptr, 4, 8, nullptr, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_reparam_kl_bwd(nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 4, 8, nullptr, nullptr, nullptr));
    REFUSED(gtg_decoder_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, 4, 8, 16, 55, nullptr, nullptr, nullptr));
    REFUSED(gtg_decoder_fwd(&f, &f, &f, &f, &f, 4, 0, 16, 55, &f, &f, nullptr));
    REFUSED(gtg_bce_loss_fwd(nullptr, nullptr, 4, 10, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_bce_loss_fwd(&f, &f, 4, 10, &f, nullptr, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_bce_loss_fwd_noalloc(nullptr, nullptr, 4, 10, nullptr, nullptr, nullptr, nullptr));
    REFUSED(gtg_bce_loss_fwd_noalloc(&f, &f, 4, 65, &f, nullptr, nullptr, nullptr));
    REFUSED(gtg_bce_loss_bwd(nullptr, nullptr, 1.f, nullptr, 4, 10, 1, nullptr, nullptr));
    REFUSED(gtg_decoder_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 10, 1.f, nullptr, 4, 8, 16, 55,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_decoder_bwd(&f, &f, &f, &f, &f, &f, &f, 10, 1.f, nullptr, 4, 8, 16, 55, &f, &f, &f, &f, &f, &f, 1 << 20,
                            nullptr));  // both dout and adj_target
    REFUSED(gtg_stage_batch(nullptr, 8, nullptr, 4, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr));
    REFUSED(gtg_gather_batch(nullptr, 0, nullptr, nullptr, 1, 10, nullptr, 1, nullptr, nullptr, 0, nullptr, nullptr, nullptr));
    REFUSED(gtg_proj_ce_fwd(nullptr, nullptr, nullptr, nullptr, 8, 16, 50, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 64,
                            GTG_MATRIX_F32, nullptr, 0, nullptr));
    REFUSED(gtg_proj_ce_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, 64, 8, 16, 50, nullptr, nullptr, nullptr,
                            GTG_MATRIX_F32, nullptr, 0, nullptr));
    REFUSED(gtg_lstm_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 2, 8, 64, 0, nullptr, nullptr, nullptr,
                         GTG_MATRIX_F16X2, nullptr, 0, nullptr));
    REFUSED(gtg_lstm_fwd(&f, &f, &f, &f, &f, nullptr, nullptr, 0, 2, 8, 64, 0, &f, &f, &f, GTG_MATRIX_F16X2, &f, 4, nullptr));
    REFUSED(gtg_lstm_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 2, 8,
                         64, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, GTG_MATRIX_F16X2, nullptr, 0, nullptr));
    REFUSED(gtg_word_drop(nullptr, 4, 2, 0, 0.3f, 0, 0, nullptr, nullptr, nullptr));
    REFUSED(gtg_word_drop(&i64, 4, 2, 0, 1.5f, 0, 0, &u8, &i64, nullptr));
    REFUSED(gtg_word_drop(&i64, 4, 2, 0, 0.3f, 0, 0, nullptr, &i64, nullptr));  // in place
    REFUSED(gtg_embed_dropout_fwd(nullptr, nullptr, 8, 10, 8, 0.f, 0, 0, nullptr, 0, nullptr, nullptr, nullptr));
    REFUSED(gtg_embed_dropout_fwd(&f, &i64, 8, 10, 6, 0.f, 0, 0, nullptr, 0, &f, nullptr, nullptr));  // width not a multiple of 4
    REFUSED(gtg_embed_dropout_fwd(&f, &i64, 8, 10, 8, 1.f, 0, 0, nullptr, 0, &f, nullptr, nullptr));  // p = 1
    REFUSED(gtg_embed_dropout_bwd(nullptr, nullptr, 8, 10, 8, 0.f, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr));
    REFUSED(gtg_embed_dropout_bwd(&f, &i64, 9, 10, 8, 0.f, 0, 0, nullptr, &f, 2, &f, 1 << 20, nullptr));  // rows % add_rows
    REFUSED(gtg_dropout(&f, 6, 0.5f, 0, 0, &f, nullptr));
    REFUSED(gtg_dropout(nullptr, 8, 0.5f, 0, 0, nullptr, nullptr));
    REFUSED(gtg_adam_step(nullptr, nullptr, nullptr, nullptr, 8, nullptr, 1e-3f, 0.5f, 0.999f, 1e-8f, 1.f, nullptr));
    REFUSED(gtg_adam_step(&f, &f, &f, &f, -1, &f, 1e-3f, 0.5f, 0.999f, 1e-8f, 1.f, nullptr));
    REFUSED(gtg_gemm_f32(&f, &f, &f, 4, 4, 4, 4, 4, 4, 1, 1, nullptr, 0, nullptr, 0, nullptr, 0, 99, nullptr));
    REFUSED(gtg_gemm_f32(&f, &f, &f, 4, 4, 4, 4, 4, 4, 1, 1, nullptr, 17, nullptr, 0, nullptr, 0, GTG_MATRIX_F32, nullptr));
    REFUSED(gtg_gemm_f32_rows(&f, &f, &f, 4, 4, 4, 4, 4, 4, 1, 1, nullptr, 0, nullptr, 0, nullptr, 99, nullptr, 0, nullptr));

    if (g_bad) {
        std::fprintf(stderr, "host check FAILED: %d of %d calls\n", g_bad, g_calls);
        return 1;
    }
    std::printf("host check ok: %d calls\n", g_calls);
    return 0;
}
