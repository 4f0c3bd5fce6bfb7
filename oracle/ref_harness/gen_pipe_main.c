/* gen_pipe_main.c -- TEST INFRASTRUCTURE: launcher of oracle/ref_harness/gen_pipe.c, which is built as a shared object because it
 * compiles the reference's slicedec.c in (calls into reference files that cannot be built here stay unbound and are never made). */
int gp_main(int argc, char **argv);
int main(int argc, char **argv) { return gp_main(argc, argv); }
