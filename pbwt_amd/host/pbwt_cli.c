/* pbwt_cli.c — `pbwt` command interpreter for the hot-path subset of the reference's CLI
 * (pbwtMain.c:276-494): a sequence of "-command args" applied in order to one current panel.
 * Supported: -check -stats -log -checkpoint -read -readSites -readAll -readMacs -write -writeSites -writeAll
 * -haps -maxWithin -longWithin -matchDynamic -siteInfo -subsample -subrange -selectSites -removeSites -buildReverse -writeReverse -readReverse.  Everything else: "not on the accelerated
 * path of this build". */
#include "pbwt_host.h"
#include <stdlib.h>
#include <string.h>

static FILE *openOrDie (const char *name, const char *what, const char *mode)
{ FILE *fp = fopen (name, mode) ;
  if (!fp) die ("failed to open %s file %s", what, name) ;
  return fp ;
}

int main (int argc, char *argv[])
{
  Panel *p = 0 ;
  FILE *fp ;
  logFile = stderr ;
  --argc ; ++argv ;
  if (!argc)
    { fprintf (stderr, "Program: pbwt (MI355X hot-path build, pbwt_amd)\nUsage: pbwt [ -<command> [options]* ]+\n"
	       "Commands: -check -stats -log <file> -checkpoint <n> -read <file> -readSites <file> -readAll <root> -readMacs <file>\n"
	       "          -write <file> -writeSites <file> -writeAll <root> -haps <file> -maxWithin -longWithin <L>\n"
	       "          -matchDynamic <file> -siteInfo <file> <kmin> <kmax> -subsample <start> <n> -subrange <start> <end>\n"
	       "          -selectSites <file> -removeSites <file>\n"
	       "          -buildReverse -writeReverse <file> -readReverse <file>\n") ;
      return 0 ;
    }
  timeUpdate (logFile) ;
  while (argc)
    { if (**argv != '-') die ("not well formed command %s\nType pbwt without arguments for help", *argv) ;
#define NEEDP if (!p) die ("%s called without a PBWT", argv[0])
      if (!strcmp (argv[0], "-check")) { isCheck = 1 ; argc -= 1 ; argv += 1 ; }
      else if (!strcmp (argv[0], "-stats")) { isStats = 1 ; argc -= 1 ; argv += 1 ; }
      else if (!strcmp (argv[0], "-log") && argc > 1)
	{ if (logFile != stderr) fclose (logFile) ; logFile = openOrDie (argv[1], "log", "w") ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-read") && argc > 1)
	{ if (p) panelDestroy (p) ; fp = openOrDie (argv[1], "read", "r") ; p = panelRead (fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-readSites") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "readSites", "r") ; panelReadSites (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-readAll") && argc > 1)
	{ if (p) panelDestroy (p) ; p = panelReadAll (argv[1]) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-checkpoint") && argc > 1)
	{ nCheckPoint = atoi (argv[1]) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-readMacs") && argc > 1)
	{ if (p) panelDestroy (p) ; fp = openOrDie (argv[1], "readMacs", "r") ; p = panelReadMacs (fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-write") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "write", "w") ; panelWrite (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-writeSites") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "writeSites", "w") ; panelWriteSites (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-writeAll") && argc > 1)
	{ NEEDP ; panelWriteAll (p, argv[1]) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-haps") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "haps", "w") ; panelWriteHaplotypes (fp, p) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-buildReverse"))
	{ NEEDP ; panelBuildReverse (p) ; argc -= 1 ; argv += 1 ; }
      else if (!strcmp (argv[0], "-writeReverse") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "writeReverse", "w") ; panelWriteReverse (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-readReverse") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "readReverse", "r") ; panelReadReverse (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-maxWithin"))
	{ NEEDP ; panelLongMatches (p, 0) ; argc -= 1 ; argv += 1 ; }
      else if (!strcmp (argv[0], "-longWithin") && argc > 1)
	{ NEEDP ; panelLongMatches (p, atoi (argv[1])) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-matchDynamic") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "matchDynamic", "r") ; panelMatchDynamic (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-siteInfo") && argc > 3)
	{ NEEDP ; fp = openOrDie (argv[1], "siteInfo", "w") ; panelSiteInfo (p, fp, atoi (argv[2]), atoi (argv[3])) ; fclose (fp) ; argc -= 4 ; argv += 4 ; }
      else if (!strcmp (argv[0], "-subsample") && argc > 2)
	{ NEEDP ; p = panelSubSampleInterval (p, atoi (argv[1]), atoi (argv[2])) ; argc -= 3 ; argv += 3 ; }
      else if (!strcmp (argv[0], "-subrange") && argc > 2)
	{ NEEDP ; p = panelSubRange (p, atoi (argv[1]), atoi (argv[2])) ; argc -= 3 ; argv += 3 ; }
      else if (!strcmp (argv[0], "-selectSites") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "selectSites", "r") ; p = panelSelectSites (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else if (!strcmp (argv[0], "-removeSites") && argc > 1)
	{ NEEDP ; fp = openOrDie (argv[1], "removeSites", "r") ; p = panelRemoveSites (p, fp) ; fclose (fp) ; argc -= 2 ; argv += 2 ; }
      else
	die ("unrecognised command %s (or missing arguments): not on the accelerated path of this build\nType pbwt without arguments for help", *argv) ;
      timeUpdate (logFile) ;
    }
  if (p) panelDestroy (p) ;
  fflush (stdout) ;
  if (logFile != stderr) fclose (logFile) ;
  return 0 ;
}
