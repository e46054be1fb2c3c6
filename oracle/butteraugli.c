#include "ce_oracle.h"
int ceo_butteraugli(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                    size_t width, size_t height, float intensity_target, double *score,
                    double *pnorm3)
{
    (void)ref; (void)ref_len; (void)test; (void)test_len; (void)width; (void)height;
    (void)intensity_target; (void)score; (void)pnorm3;
    return CEO_BACKEND; /* placeholder until the restatement lands */
}
