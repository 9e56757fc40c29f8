/* exitguard -- a bound on process teardown for bench.py (test infrastructure, not part of the product library).
 * Round 4 saw one full bench run in dozens sit in interpreter / runtime teardown for minutes AFTER its result line was out; the script then
 * left through os._exit, which hid the teardown instead of exercising it.  Now every run tears down normally (clouds, lh_destroy, interpreter
 * finalisation, the HIP runtime's own atexit), and this guard -- a detached native thread that needs neither the GIL nor a living interpreter --
 * ends the process if that takes longer than the stated number of seconds, saying on stderr which phase was the last to be announced -- with
 * the run's own exit code when that was a failure already, with 97 when the run itself had succeeded: a teardown that hangs is a failure a
 * driver must be able to see in the exit status.     gcc -O2 -shared -fPIC -o libexitguard.so exitguard.c -lpthread */
#include <pthread.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static int g_seconds, g_code;
static char g_phase[128] = "armed";

void exitguard_phase(const char* name) {
  strncpy(g_phase, name ? name : "", sizeof g_phase - 1);
  g_phase[sizeof g_phase - 1] = 0;
}
static void* guard(void* unused) {
  (void)unused;
  sleep((unsigned)g_seconds);
  char msg[256];
  const int code = g_code != 0 ? g_code : 97;
  int n = snprintf(msg, sizeof msg, "[exitguard] teardown still running after %d s (last phase: %s): leaving with exit code %d\n", g_seconds, g_phase, code);
  if (n > 0) (void)!write(2, msg, (size_t)n);
  _exit(code);
  return NULL;
}
int exitguard_arm(int seconds, int code) {
  g_seconds = seconds;
  g_code = code;
  pthread_t t;
  if (pthread_create(&t, NULL, guard, NULL) != 0) return -1;
  pthread_detach(t);
  return 0;
}
