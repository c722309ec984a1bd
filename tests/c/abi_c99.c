#include <tm_hip.h>
#include <stdio.h>
int main(void) {
    tm_ctx* ctx = 0;
    int rc = tm_ctx_create(0, &ctx);
    printf("%d %s\n", rc, tm_last_error());
    return 0;
}
